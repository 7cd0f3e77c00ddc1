#!/bin/bash
mkdir -p gpurun_out/r6
echo "== in-situ tuning of the one-frame program on four lanes, lone-frame latency guarded"; timeout 2400 python scripts/tune_in_situ.py --batch 1 --shapes 60 --alts 4 --seconds 1500 gpurun_out/r6/tuning_insitu_b1.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/tune_in_situ_b1.txt | grep -v "^  M=" | tail -30
grep -c "^  M=" gpurun_out/r6/tune_in_situ_b1.txt; grep "KEPT" gpurun_out/r6/tune_in_situ_b1.txt | head -20
echo "== before"; timeout 600 python scripts/slots_sweep.py 1x4 1x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_before_insitu_b1.txt; timeout 300 python scripts/lone_frame.py --tag before 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
echo "== after"; VSD_TUNING=gpurun_out/r6/tuning_insitu_b1.json timeout 600 python scripts/slots_sweep.py 1x4 1x3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_insitu_b1.txt; VSD_TUNING=gpurun_out/r6/tuning_insitu_b1.json timeout 300 python scripts/lone_frame.py --tag after 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
