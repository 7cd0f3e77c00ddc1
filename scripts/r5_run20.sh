#!/bin/bash
# layer tables of the final build, workgroup time of the timed program (fresh instrumented build), one-frame launches in throughput-mode forms
mkdir -p gpurun_out/r5
echo "== layer tables"; bash scripts/layer_table.sh round5f_b1 1 2>&1 | tail -3 | cut -c1-300; bash scripts/layer_table.sh round5f_b5 5 2>&1 | tail -3 | cut -c1-300
echo "== CU time"; VSD_LIB=videosd_amd/libvsd_tl.so timeout 900 python scripts/wg_cu_time.py --seconds 2.0 --out gpurun_out/r5/wg_cu_time_5x4.txt 2>&1 | grep -v amdgpu.ids | grep -v '^{"runs"' | tail -7 | cut -c1-330
echo "== lone frame, latency forms"; timeout 900 python scripts/lone_frame.py --tag mode0 --lanes 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('p50_ms','gpu_p50_ms','serial_ms','fps_1x3','fps_1x4','prepare_s')})"
echo "== lone frame, throughput-mode forms"; timeout 1500 python scripts/lone_frame.py --tag mode1 --lanes --mode1 --save-tuning gpurun_out/r5/tuning_mode1_b1.json 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('p50_ms','gpu_p50_ms','serial_ms','fps_1x3','fps_1x4','prepare_s')})"
