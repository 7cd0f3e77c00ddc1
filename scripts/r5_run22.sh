#!/bin/bash
# pipeline 8 (fragments half a step ahead): parity, then the tuner's tables with it among the candidates -- small 1x1 GEMMs of the one-frame
# program, and a few 5-frame shapes alone / with four lanes busy
mkdir -p gpurun_out/r5
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pipelines_are_bit or all_tiles_splitk or splitk or order" 2>&1 | tail -3
echo "== option fuzz (table mode)"; timeout 300 python scripts/option_fuzz.py 120 3 table 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r5/robustness_option_fuzz.txt
echo "== small GEMM candidates"; timeout 1200 python scripts/small_gemm_candidates.py > gpurun_out/r5/small_gemm_candidates_p8.txt 2>/dev/null; grep -A3 "^M=" gpurun_out/r5/small_gemm_candidates_p8.txt | cut -c1-160 | head -90
for shape in "5 32 32 640 640 1" "5 64 64 320 320 1" "5 16 16 1280 1280 1" "5 32 32 640 640 3" "5 64 64 320 320 3"; do
  for mode in "" "--mode1"; do
    echo "== $shape $mode"; timeout 300 python scripts/tune_conv_shape.py $shape $mode 2>&1 | grep -v amdgpu.ids | head -8
  done
done
