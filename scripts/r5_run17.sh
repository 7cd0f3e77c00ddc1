#!/bin/bash
# final measurements of the round-5 build: table completed (group / pair entries), default bench line, workgroup time of the timed
# program, the rocprofv3 set (kernel stats of the bench command, layer tables, PMC)
mkdir -p gpurun_out/r5
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "group or twin or pair or splitk or epilogue or own or groupnorm or halo" 2>&1 | tail -4
echo "== lone frame"; timeout 900 python scripts/lone_frame.py --tag gn_reducer_prefetch --lanes 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-330
echo "== table"; timeout 1500 python scripts/update_tuning.py 2>&1 | grep -v amdgpu.ids | tail -4; cp profiles/tuning_mi355x.json gpurun_out/r5/tuning_mi355x.json
echo "== bench"; timeout 1200 python bench.py > gpurun_out/r5/bench_final.json 2> gpurun_out/r5/bench_final.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5/bench_final.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms","fps_one_frame_per_launch","fps_end_to_end","fps_without_controlnet","api_fps","prepare_ms")})
print(d.get("fps_by_frames_per_launch_x_launches_in_flight"), d["roofline"]["achieved"], d["config"].get("kernel_launches_by_form"))
PY
tail -2 gpurun_out/r5/bench_final.err
echo "== CU time"; VSD_LIB=videosd_amd/libvsd_tl.so timeout 900 python scripts/wg_cu_time.py --seconds 2.0 --out gpurun_out/r5/wg_cu_time_5x4.txt 2>&1 | grep -v amdgpu.ids | grep -v '^{"runs"' | tail -16 | cut -c1-330
echo "== rocprofv3 set"; timeout 1500 bash scripts/collect_profiles.sh round5f 5 2>&1 | grep -v amdgpu.ids | tail -20 | cut -c1-300
