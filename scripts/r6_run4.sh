#!/bin/bash
mkdir -p gpurun_out/r6
echo "== bench at HEAD (shipped table)"; timeout 900 python bench.py 2>gpurun_out/r6/bench_head.err | tail -1 > gpurun_out/r6/bench_head.json; head -c 600 gpurun_out/r6/bench_head.json; echo
echo "== full gpu suite"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r6/gpu_suite_head.txt
echo "== retune throughput-mode entries with the eight-wave / 256x256 candidates"; VSD_RETUNE_SECONDS=1000 timeout 1500 python scripts/retune_mode1.py gpurun_out/r6/tuning_w8.json 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r6/retune_mode1.txt
echo "== sweep, shipped table"; timeout 600 python scripts/slots_sweep.py 5x4 8x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_shipped.txt
echo "== sweep, re-timed table"; VSD_TUNING=gpurun_out/r6/tuning_w8.json timeout 600 python scripts/slots_sweep.py 5x4 8x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_w8.txt
