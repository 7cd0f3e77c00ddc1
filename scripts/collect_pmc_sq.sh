#!/bin/bash
# SQ wait-state / LDS bank-conflict counters per kernel family: two --pmc passes (kernel-trace only), then the summary.
# usage (on the GPU box): TAG=round2 NB=5 [EXTRA=--lanes] bash scripts/collect_pmc_sq.sh   (NB = frames per launch; EXTRA=--lanes: the throughput-mode forms)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_a /tmp/pmc_b
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pmc_a -- python3 scripts/profile_frame.py --batch=${NB:-5} ${EXTRA:-} sq_a > /tmp/pmc_a.log 2>&1; tail -2 /tmp/pmc_a.log | cut -c1-200
timeout 500 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_b -- python3 scripts/profile_frame.py --batch=${NB:-5} ${EXTRA:-} sq_b > /tmp/pmc_b.log 2>&1; tail -2 /tmp/pmc_b.log | cut -c1-200
ls /tmp/pmc_a/*/ /tmp/pmc_b/*/ | head
N=0  # the last eager frame (from its preprocess_rgb dispatch on)
python3 scripts/pmc_sq_summary.py "gpurun_out/${TAG:-round3}_pmc_sq.json" "$N" /tmp/pmc_a /tmp/pmc_b | cut -c1-2500
