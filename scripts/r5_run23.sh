#!/bin/bash
# SQ counters per kernel family and a lone frame's kernel timeline on the final build; the guard-page fuzz test and smoke on the rebuilt library
mkdir -p gpurun_out/r5
echo "== smoke + guard page test"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "unmapped or outside" 2>&1 | tail -2
echo "== SQ counters"; TAG=round5f NB=5 bash scripts/collect_pmc_sq.sh 2>&1 | tail -4 | cut -c1-1500
echo "== lone frame timeline"; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; rm -rf /tmp/ft; rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -- python3 scripts/lone_frame.py --tag trace > /tmp/ft.log 2>&1; python3 scripts/frame_timeline.py /tmp/ft/*/*_kernel_trace.csv > gpurun_out/r5/frame_timeline.txt 2>&1; head -12 gpurun_out/r5/frame_timeline.txt | cut -c1-300
