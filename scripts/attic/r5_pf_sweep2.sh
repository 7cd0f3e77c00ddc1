#!/bin/bash
# round 5: the weight prefetcher on compute units of its own (VSD_POOL_MASK=p<k>: the launch streams give up k CUs per XCD)
mkdir -p gpurun_out/r5
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['tag'], 'p50', d['p50_ms'], 'gpu', d['gpu_p50_ms'], 'serial', d['serial_ms'], 'pf', d['prefetch'])"; }
run() { tag=$1; shift; env "$@" timeout 300 python scripts/lone_frame.py --tag $tag 2>&1 | tail -1 | show; }
runno() { tag=$1; shift; env "$@" timeout 300 python scripts/lone_frame.py --tag $tag --no-prefetch 2>&1 | tail -1 | show; }
runno nopf_full
runno nopf_p1 VSD_POOL_MASK=p1
runno nopf_p2 VSD_POOL_MASK=p2
run p1_wg8 VSD_POOL_MASK=p1 VSD_PF_WGS=8
run p1_wg16 VSD_POOL_MASK=p1 VSD_PF_WGS=16
run p2_wg16 VSD_POOL_MASK=p2 VSD_PF_WGS=16
run p2_wg32 VSD_POOL_MASK=p2 VSD_PF_WGS=32
run p2_wg32_la16 VSD_POOL_MASK=p2 VSD_PF_WGS=32 VSD_PF_LOOKAHEAD_MB=16
run p2_wg32_la96 VSD_POOL_MASK=p2 VSD_PF_WGS=32 VSD_PF_LOOKAHEAD_MB=96
run p4_wg64 VSD_POOL_MASK=p4 VSD_PF_WGS=64
