#!/bin/bash
mkdir -p gpurun_out/r6
echo "== eight-wave parity"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "eight_wave or 256x128 or all_tiles_splitk or geglu or softmax" 2>&1 | tail -8
echo "== w8 probe lanes"; timeout 1500 python scripts/w8_probe.py --mode1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/w8_probe2_mode1.txt
echo "== w8 probe alone"; timeout 1200 python scripts/w8_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/w8_probe2_mode0.txt
echo "== dispatch ceiling on this host"; nproc; for w in 8 4; do timeout 300 python scripts/dispatch_ceiling.py --workers $w 2>&1 | tail -1 | tee -a gpurun_out/r6/dispatch_ceiling_gpu_host.txt; done
