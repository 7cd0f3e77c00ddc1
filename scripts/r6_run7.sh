#!/bin/bash
mkdir -p gpurun_out/r6
echo "== full gpu suite"; timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r6/gpu_suite_c64.txt
echo "== profiles"; timeout 1500 bash scripts/collect_profiles.sh round6 5 2>&1 | tail -30
