#!/bin/bash
mkdir -p gpurun_out/r6
echo "== persistent 64-channel conv: parity"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent_64 or halo_patch" 2>&1 | tail -8
echo "== re-time the 64-channel entries"; VSD_RETUNE_SECONDS=900 timeout 1500 python scripts/retune_c64.py gpurun_out/r6/tuning_c64.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/retune_c64.txt | tail -150
echo "== sweep, table before"; timeout 600 python scripts/slots_sweep.py 5x4 1x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_before_c64.txt
echo "== sweep, table with the persistent form"; VSD_TUNING=gpurun_out/r6/tuning_c64.json timeout 600 python scripts/slots_sweep.py 5x4 1x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_c64.txt
