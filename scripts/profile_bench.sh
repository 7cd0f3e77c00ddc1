#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box); writes the judged summary into gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r1}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_under_rocprof_${TAG}.json 2>/tmp/err.log
python scripts/shorten_stats.py /tmp/prof_bench/*/*_kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv
tail -c 700 gpurun_out/bench_under_rocprof_${TAG}.json
head -14 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-220
