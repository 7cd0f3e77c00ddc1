#!/bin/bash
# lock step at every batch size (throughput-mode pairs in their members' own form): table completed, whole GPU suite, default bench line
mkdir -p gpurun_out/r5
echo "== table"; timeout 1500 python scripts/update_tuning.py 2>&1 | grep -v amdgpu.ids | tail -2; cp profiles/tuning_mi355x.json gpurun_out/r5/tuning_mi355x_final.json
echo "== gpu suite"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== bench"; timeout 1200 python bench.py > gpurun_out/r5/bench_final4.json 2> gpurun_out/r5/bench_final4.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/bench_final4.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms","p50_latency_ms_under_load","fps_one_frame_per_launch","fps_end_to_end","fps_without_controlnet","api_fps","api_fps_one_at_a_time","prepare_ms")})
print(d.get("fps_by_frames_per_launch_x_launches_in_flight"), d["roofline"]["achieved"], d["roofline"]["frac"], d["config"].get("kernel_launches_by_form"), d.get("parity"), d["cpu_baseline"]["sample"])
PY
echo "== bench under the kernel tracer"; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/prof_bench; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-api > gpurun_out/round5f_bench_under_rocprof.json 2>/tmp/err_bench.log; python3 scripts/shorten_stats.py /tmp/prof_bench/*/*_kernel_stats.csv gpurun_out/round5f_bench_kernel_stats.csv; head -4 gpurun_out/round5f_bench_kernel_stats.csv | cut -c1-160
