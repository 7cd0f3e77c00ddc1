#!/bin/bash
# fused tails at every tile height (VSD_TAIL_BM forces one; the default picks by rounds x round time, csrc/fused_tail.hip)
for bm in 16 32 48 64 80; do echo "== VSD_TAIL_BM=$bm"; VSD_TAIL_BM=$bm python3 scripts/tail_bench.py ${@:-4096 8192 12288 16384 20480} 2>&1 | grep -v amdgpu | tail -5; done
echo "== default choice"; python3 scripts/tail_bench.py ${@:-4096 8192 12288 16384 20480} 2>&1 | grep -v amdgpu | tail -5
