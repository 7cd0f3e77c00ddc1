#!/bin/bash
mkdir -p gpurun_out/r6
echo "== eight-wave parity + rccl"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_rccl_one_gpu.py -x -q -m gpu -k "eight_wave or rccl or 256x128" 2>&1 | tail -15
echo "== w8 probe alone"; timeout 1200 python scripts/w8_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/w8_probe_mode0.txt
echo "== w8 probe lanes"; timeout 1500 python scripts/w8_probe.py --mode1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/w8_probe_mode1.txt
