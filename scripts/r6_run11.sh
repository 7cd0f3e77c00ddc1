#!/bin/bash
mkdir -p gpurun_out/r6
echo "== layer tables"; 
bash scripts/layer_table.sh round6_b1 1
bash scripts/layer_table.sh round6_b5 5
bash scripts/layer_table.sh round6_b5_lanes 5 --lanes
echo "== SQ counters, alone-tuned forms"; TAG=round6 NB=5 bash scripts/collect_pmc_sq.sh 2>&1 | tail -1 | cut -c1-300
