#!/bin/bash
# round 5, GPU call 5: attention A/B, op tests, CU-time of the timed program, stress-weights fixture + test, one-frame retune
mkdir -p gpurun_out/r5
echo "== attention, new build"; ATTN_BENCH_SHORT=1 timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/attn_new.txt
echo "== attention, old build"; ATTN_BENCH_SHORT=1 ATTN_BENCH_LIB=videosd_amd/libvsd_attn_old.so timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/attn_old.txt
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -4
echo "== CU time"; VSD_LIB=videosd_amd/libvsd_tl.so timeout 900 python scripts/wg_cu_time.py --out gpurun_out/r5/wg_cu_time_5x4.txt 2>&1 | grep -v amdgpu.ids | tail -22 | cut -c1-400
echo "== stress golden"; cp tests/golden/fullsize_oracle.npz gpurun_out/r5/fullsize_oracle.npz; timeout 1500 python scripts/make_fullsize_golden.py gpurun_out/r5/fullsize_oracle.npz --only stress512 2>&1 | grep -v amdgpu.ids | tail -3
cp gpurun_out/r5/fullsize_oracle.npz tests/golden/fullsize_oracle.npz
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -k "range_stress" 2>&1 | tail -12
echo "== one-frame retune"; timeout 900 python scripts/lone_frame.py --tag e1_retune --retune --lanes --save-tuning gpurun_out/r5/tuning_e1.json 2>&1 | tail -1 | cut -c1-900
