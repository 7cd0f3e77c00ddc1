#!/bin/bash
# the round's last build (grouped kernel without its scratch copy, fused GroupNorm without spills, group entries timed again): whole GPU
# suite, default bench line, the rocprofv3 set
mkdir -p gpurun_out/r5
echo "== gpu suite"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bench"; timeout 1200 python bench.py > gpurun_out/r5/bench_final3.json 2> gpurun_out/r5/bench_final3.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/bench_final3.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms","p50_latency_ms_under_load","fps_one_frame_per_launch","fps_end_to_end","fps_without_controlnet","api_fps","api_fps_one_at_a_time","prepare_ms")})
print(d.get("fps_by_frames_per_launch_x_launches_in_flight"), d["roofline"]["achieved"], d["roofline"]["frac"], d["config"].get("kernel_launches_by_form"), d.get("parity"), d["cpu_baseline"]["sample"])
PY
echo "== rocprofv3 set"; timeout 1500 bash scripts/collect_profiles.sh round5f 5 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-260
