#!/bin/bash
bash scripts/layer_table.sh round6f_b1 1
bash scripts/layer_table.sh round6f_b5 5
bash scripts/layer_table.sh round6f_b5_lanes 5 --lanes
TAG=round6f NB=5 bash scripts/collect_pmc_sq.sh 2>&1 | tail -1 | cut -c1-200; TAG=round6f_lanes NB=5 EXTRA=--lanes bash scripts/collect_pmc_sq.sh 2>&1 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 scripts/profile_frame.py --batch=5 round6f_pmc > /tmp/pmc_$C.log 2>&1
done
python3 scripts/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/ops_round6f_pmc.json gpurun_out/round6f_pmc_conv_gemm.json | cut -c1-300
