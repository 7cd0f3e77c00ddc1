#!/bin/bash
# tail_b on 64- against 80-token tiles (VSD_TAIL_BM forces one; default picks by rounds x height)
for bm in 64 80; do echo "== VSD_TAIL_BM=$bm"; VSD_TAIL_BM=$bm python3 scripts/tail_bench.py 20480 16384 12288 32768 2>&1 | grep -v amdgpu | tail -8; done
