#!/bin/bash
# Everything profiles/ holds for a round, collected on the GPU box (gpurun): rocprofv3 kernel-trace summary of the bench
# command, per-layer tables of one eager pass (1 and 3 frames per launch), fabric-side PMC bytes of the conv kernel.
# usage: scripts/collect_profiles.sh <tag> [frames per launch of the batched tables / PMC passes, default 5]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r1}
NB=${2:-5}
mkdir -p gpurun_out
# 1. the bench command under the kernel tracer
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-api > gpurun_out/${TAG}_bench_under_rocprof.json 2>/tmp/err_bench.log
python3 scripts/shorten_stats.py /tmp/prof_bench/*/*_kernel_stats.csv gpurun_out/${TAG}_bench_kernel_stats.csv
tail -c 600 gpurun_out/${TAG}_bench_under_rocprof.json; echo
head -8 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-160
# 2. per-layer tables
for B in 1 $NB; do
  rm -rf /tmp/pf$B
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pf$B -- python3 scripts/profile_frame.py --batch=$B ${TAG}_b$B > /tmp/pf$B.log 2>&1
  python3 scripts/analyze_trace.py /tmp/pf$B/*/*_kernel_trace.csv gpurun_out/ops_${TAG}_b$B.json > gpurun_out/${TAG}_layer_table_b$B.txt
  head -2 gpurun_out/${TAG}_layer_table_b$B.txt
done
# 3. PMC: fabric-side bytes of the conv kernel, one counter per pass (no tracing options besides kernel-trace)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 scripts/profile_frame.py --batch=$NB ${TAG}_pmc > /tmp/pmc_$C.log 2>&1
done
python3 scripts/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/ops_${TAG}_pmc.json gpurun_out/${TAG}_pmc_conv_gemm.json
