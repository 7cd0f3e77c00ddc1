#!/bin/bash
mkdir -p gpurun_out/r5
echo "== CU time"; VSD_LIB=videosd_amd/libvsd_tl.so timeout 900 python scripts/wg_cu_time.py --seconds 2.0 --out gpurun_out/r5/wg_cu_time_5x4.txt 2>&1 | grep -v amdgpu.ids | grep -v '^{"runs"' | tail -16 | cut -c1-330
echo "== stress golden (fp32 + fp16-storage emulation)"; cp tests/golden/fullsize_oracle.npz gpurun_out/r5/fullsize_oracle.npz; timeout 1500 python scripts/make_fullsize_golden.py gpurun_out/r5/fullsize_oracle.npz --only stress512 2>&1 | grep -v amdgpu.ids | tail -4
