#!/bin/bash
# the lean pointwise form (pipeline 8): its parity tests, then every tuner candidate of the small 1x1 GEMMs with it among them
mkdir -p gpurun_out/r5
echo "== lean tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "lean" 2>&1 | tail -15
echo "== candidates"; timeout 1200 python scripts/small_gemm_candidates.py > gpurun_out/r5/small_gemm_candidates_lean.txt 2> gpurun_out/r5/small_gemm_candidates_lean.err; grep -A4 "^M=" gpurun_out/r5/small_gemm_candidates_lean.txt | cut -c1-200; tail -3 gpurun_out/r5/small_gemm_candidates_lean.err
