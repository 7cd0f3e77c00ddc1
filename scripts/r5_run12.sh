#!/bin/bash
mkdir -p gpurun_out/r5
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms","fps_one_frame_per_launch","fps_end_to_end","fps_without_controlnet")})
print(d.get("fps_by_frames_per_launch_x_launches_in_flight"))
PY
}
echo "== guard"; timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -k "unmapped or same_bits" 2>&1 | tail -3
echo "== bench twin"; timeout 1200 python bench.py > gpurun_out/r5/bench_twin.json 2> gpurun_out/r5/bench_twin.err; show gpurun_out/r5/bench_twin.json
echo "== bench VSD_NO_TWIN"; VSD_NO_TWIN=1 timeout 1200 python bench.py > gpurun_out/r5/bench_notwin.json 2> gpurun_out/r5/bench_notwin.err; show gpurun_out/r5/bench_notwin.json
echo "== bench twin again"; timeout 1200 python bench.py > gpurun_out/r5/bench_twin2.json 2> gpurun_out/r5/bench_twin2.err; show gpurun_out/r5/bench_twin2.json
