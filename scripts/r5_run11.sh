#!/bin/bash
mkdir -p gpurun_out/r5
echo "== new tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv_group or twin or pair or splitk" 2>&1 | tail -8
echo "== lone frame (two forms)"; timeout 900 python scripts/lone_frame.py --tag twoforms --lanes --save-tuning gpurun_out/r5/tuning_twin.json 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-1200
echo "== pipeline parity"; timeout 1500 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -4
echo "== bench"; timeout 1200 python bench.py > gpurun_out/r5/bench_twin.json 2> gpurun_out/r5/bench_twin.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/bench_twin.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms","fps_one_frame_per_launch","fps_end_to_end","fps_without_controlnet")})
print(d.get("fps_by_frames_per_launch_x_launches_in_flight"))
print(d["roofline"]["achieved"], d["roofline"]["launches_per_pass"], d["config"]["kernel_launches_per_graph_replay"], d["config"]["kernel_launches_per_single_frame_graph"])
PY
tail -3 gpurun_out/r5/bench_twin.err
