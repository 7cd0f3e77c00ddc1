#!/bin/bash
# round 5: lone-frame latency with the weight prefetcher at several settings (one JSON line each)
mkdir -p gpurun_out/r5
run() { tag=$1; shift; env "$@" timeout 300 python scripts/lone_frame.py --tag $tag 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['tag'], 'p50', d['p50_ms'], 'gpu', d['gpu_p50_ms'], 'pf', d['prefetch'])"; }
timeout 300 python scripts/lone_frame.py --tag nopf --no-prefetch 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['tag'], 'p50', d['p50_ms'], 'gpu', d['gpu_p50_ms'])"
run wg16 VSD_PF_WGS=16
run wg32 VSD_PF_WGS=32
run wg64 VSD_PF_WGS=64
run wg32_nt VSD_PF_WGS=32 VSD_PF_NT=1
run wg32_la16 VSD_PF_WGS=32 VSD_PF_LOOKAHEAD_MB=16
run wg32_la128 VSD_PF_WGS=32 VSD_PF_LOOKAHEAD_MB=128
run wg32_big VSD_PF_WGS=32 VSD_PF_MIN_ENTRY_KB=4096
run wg32_lead8 VSD_PF_WGS=32 VSD_PF_MIN_LEAD_KB=8192
run wg128 VSD_PF_WGS=128
