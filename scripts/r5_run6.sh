#!/bin/bash
mkdir -p gpurun_out/r5
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -3
echo "== CU time (debug)"; VSD_CUT_DEBUG=1 VSD_LIB=videosd_amd/libvsd_tl.so timeout 900 python scripts/wg_cu_time.py --seconds 1.0 --out gpurun_out/r5/wg_cu_time_5x4.txt 2>&1 | grep -v amdgpu.ids | grep -v '^{"runs"' | tail -34 | cut -c1-330
echo "== stress: first non-finite"; timeout 600 python scripts/find_nonfinite.py --stress 2>&1 | grep -v amdgpu.ids | tail -18 | cut -c1-250
echo "== plain: headroom"; timeout 600 python scripts/find_nonfinite.py 2>&1 | grep -v amdgpu.ids | tail -15 | cut -c1-250
