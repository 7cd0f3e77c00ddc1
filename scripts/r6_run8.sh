#!/bin/bash
mkdir -p gpurun_out/r6
echo "== in-situ tuning at 5 x 4"; timeout 2400 python scripts/tune_in_situ.py --batch 5 --shapes 45 --alts 3 --seconds 1700 gpurun_out/r6/tuning_insitu.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/tune_in_situ.txt | tail -70
echo "== sweep, table before"; timeout 600 python scripts/slots_sweep.py 5x4 8x4 3x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_before_insitu.txt
echo "== sweep, in-situ table"; VSD_TUNING=gpurun_out/r6/tuning_insitu.json timeout 600 python scripts/slots_sweep.py 5x4 8x4 3x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_insitu.txt
