#!/bin/bash
mkdir -p gpurun_out/r6
echo "== configs[1] through the C host"; timeout 1200 python scripts/plan_bench.py --dir /tmp 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/plan_bench.txt
