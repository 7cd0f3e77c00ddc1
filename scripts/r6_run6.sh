#!/bin/bash
mkdir -p gpurun_out/r6
echo "== parity: persistent 64-channel conv, groupnorm (cooperative form)"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent_64 or halo_patch or groupnorm or pair_runs" 2>&1 | tail -8
echo "== groupnorm timings, cooperative form on | off"
for B in 1 5; do for c in 1 0; do echo "B=$B VSD_GN_COOP=$c"; VSD_GN_COOP=$c timeout 300 python scripts/gn_bench.py $B 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/r6/gn_coop_bench.txt
echo "== re-time the 64-channel entries"; VSD_RETUNE_SECONDS=900 timeout 1500 python scripts/retune_c64.py gpurun_out/r6/tuning_c64.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/retune_c64.txt; tail -2 gpurun_out/r6/retune_c64.txt
for c in 0 1; do
echo "== sweep, VSD_GN_COOP=$c, table with the persistent form"; VSD_GN_COOP=$c VSD_TUNING=gpurun_out/r6/tuning_c64.json timeout 600 python scripts/slots_sweep.py 5x4 1x4 3x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_c64_coop$c.txt
echo "== lone frame, VSD_GN_COOP=$c"; VSD_GN_COOP=$c VSD_TUNING=gpurun_out/r6/tuning_c64.json timeout 600 python scripts/lone_frame.py --tag coop$c 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r6/lone_coop$c.json
done
