#!/bin/bash
mkdir -p gpurun_out/r6
echo "== attention parity"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or pair_runs" 2>&1 | tail -6
echo "== attention timings: the lazy form | the plain form"; for off in "" 1; do echo "VSD_ATTN_NO_LAZY=$off"; ( [ -n "$off" ] && export VSD_ATTN_NO_LAZY=1; ATTN_BENCH_SHORT=1 timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-200 ); done | tee gpurun_out/r6/attn_lazy_bench.txt
echo "== sweep, lazy form"; timeout 600 python scripts/slots_sweep.py 5x4 1x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_attn_lazy.txt
echo "== sweep, plain form"; VSD_ATTN_NO_LAZY=1 timeout 600 python scripts/slots_sweep.py 5x4 1x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_attn_plain.txt
echo "== pipeline parity"; timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "mini_pipeline or baseline_config2_512_four or same_bits" 2>&1 | tail -4
