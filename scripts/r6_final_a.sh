#!/bin/bash
# round 6, final set, part A: everything bench.py's line quotes from profiles/ (layer tables, PMC, workgroup time) + the GPU suite
mkdir -p gpurun_out/r6f
echo "== bench under the kernel tracer, layer tables, fabric bytes"; timeout 1500 bash scripts/collect_profiles.sh round6f 5 2>&1 | tail -12 | cut -c1-300
echo "== layer table of the throughput-mode forms"; bash scripts/layer_table.sh round6f_b5_lanes 5 --lanes
echo "== SQ counters"; TAG=round6f NB=5 bash scripts/collect_pmc_sq.sh 2>&1 | tail -1 | cut -c1-200; TAG=round6f_lanes NB=5 EXTRA=--lanes bash scripts/collect_pmc_sq.sh 2>&1 | tail -1 | cut -c1-200
echo "== workgroup time of the timed program"; VSD_LIB=videosd_amd/libvsd_tl.so timeout 900 python scripts/wg_cu_time.py --out gpurun_out/r6f/wg_cu_time_5x4.txt 2>&1 | grep -v amdgpu.ids | tail -30 | cut -c1-200
echo "== the other configs"; timeout 900 python scripts/bench_configs.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r6f/configs.txt | cut -c1-300
