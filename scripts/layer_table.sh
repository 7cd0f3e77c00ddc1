#!/bin/bash
# per-layer table of one eager 512x512 4-step pass under the kernel tracer: scripts/layer_table.sh <tag> <batch> [extra profile_frame args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; B=$2; shift; shift
rm -rf /tmp/pf_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_$TAG -- python3 scripts/profile_frame.py --batch=$B "$@" ${TAG} > /tmp/pf_$TAG.log 2>&1
python3 scripts/analyze_trace.py /tmp/pf_$TAG/*/*_kernel_trace.csv gpurun_out/ops_${TAG}.json > gpurun_out/${TAG}_layer_table.txt
head -2 gpurun_out/${TAG}_layer_table.txt
