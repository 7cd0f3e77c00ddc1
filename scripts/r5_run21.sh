#!/bin/bash
# robustness runs on the final build (grouped launches now among the guard-page cases), then the final default bench line
mkdir -p gpurun_out/r5
( echo "== guard_page_fuzz 150 s seed 5"; timeout 400 python scripts/guard_page_fuzz.py 150 5 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== guard_page_fuzz 150 s seed 6"; timeout 400 python scripts/guard_page_fuzz.py 150 6 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== option_fuzz 150 s"; timeout 400 python scripts/option_fuzz.py 150 3 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== guard_fuzz 90 s"; timeout 300 python scripts/guard_fuzz.py 90 3 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== guard_page_engine"; timeout 600 python scripts/guard_page_engine.py 2>&1 | grep -v amdgpu.ids | tail -3 ) | tee gpurun_out/r5/robustness_runs.txt | cut -c1-400
echo "== bench"; timeout 1200 python bench.py > gpurun_out/r5/bench_final2.json 2> gpurun_out/r5/bench_final2.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/bench_final2.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms","fps_one_frame_per_launch","fps_end_to_end","fps_without_controlnet","api_fps","api_fps_one_at_a_time")})
print(d.get("fps_by_frames_per_launch_x_launches_in_flight"), d["roofline"]["achieved"], d["roofline"]["in_situ"]["source"][:60])
PY
