#!/bin/bash
# round 6, final set, part B: the GPU suite and the default bench line on the final build (profiles/ holds part A's tables)
mkdir -p gpurun_out/r6f
echo "== full gpu suite"; timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r6f/gpu_suite.txt
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/r6f/bench.err | tail -1 > gpurun_out/r6f/bench.json; head -c 400 gpurun_out/r6f/bench.json; echo
echo "== C host on four lanes"; timeout 900 python scripts/plan_bench.py --dir /tmp 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6f/plan_bench.txt | cut -c1-700
