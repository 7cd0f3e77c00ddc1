#!/bin/bash
# re-entry check of the twin-encoder build: the whole GPU suite, then the default bench line (twice: boxes differ) and the one-stream form off
mkdir -p gpurun_out/r5
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms","fps_one_frame_per_launch","fps_end_to_end","fps_without_controlnet","api_fps")})
print(d.get("fps_by_frames_per_launch_x_launches_in_flight"))
print(d["roofline"]["achieved"], d["roofline"].get("launches_per_pass"), d["config"].get("kernel_launches_per_graph_replay"), d["config"].get("kernel_launches_per_single_frame_graph"))
PY
}
echo "== gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== bench"; timeout 900 python bench.py > gpurun_out/r5/bench_a.json 2> gpurun_out/r5/bench_a.err; show gpurun_out/r5/bench_a.json; tail -2 gpurun_out/r5/bench_a.err
echo "== bench VSD_NO_TWIN"; VSD_NO_TWIN=1 timeout 900 python bench.py > gpurun_out/r5/bench_notwin.json 2> gpurun_out/r5/bench_notwin.err; show gpurun_out/r5/bench_notwin.json
