#!/bin/bash
# usage: scripts/kprof.sh <tag> <script.py> [args...]   -- rocprofv3 kernel-trace + stats of a python script; prints the top rows
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
rm -rf /tmp/kprof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof_$TAG -- python3 "$@" > gpurun_out/kprof_${TAG}.log 2>/tmp/kprof_err.log
python3 scripts/shorten_stats.py /tmp/kprof_$TAG/*/*_kernel_stats.csv gpurun_out/kprof_${TAG}_stats.csv
grep -v amdgpu.ids gpurun_out/kprof_${TAG}.log | tail -40
head -${KPROF_ROWS:-16} gpurun_out/kprof_${TAG}_stats.csv | cut -c1-200
