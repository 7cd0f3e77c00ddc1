#!/bin/bash
mkdir -p gpurun_out/r6
echo "== parity: persistent conv incl. thin output"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "persistent_64 or halo_patch or guard" 2>&1 | tail -4
echo "== probe alone"; timeout 600 python scripts/c64_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/c64_probe_alone.txt
echo "== probe four lanes"; timeout 900 python scripts/c64_probe.py --mode1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/c64_probe_lanes.txt
echo "== re-time the 64-input-channel entries (thin outputs too)"; VSD_RETUNE_SECONDS=900 timeout 1500 python scripts/retune_c64.py gpurun_out/r6/tuning_c64b.json 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/retune_c64b.txt; tail -1 gpurun_out/r6/retune_c64b.txt; grep " 3, 576\|M=.*-> (5, 1, False, 10)" gpurun_out/r6/retune_c64b.txt | head -5
echo "== clocks under load"; timeout 300 bash scripts/clock_probe.sh; head -3 gpurun_out/r6/clock_probe.txt; tail -3 gpurun_out/r6/clock_probe.txt; cat gpurun_out/r6/clock_probe_sweep.txt | grep -v amdgpu.ids
echo "== sweep with the new table"; VSD_TUNING=gpurun_out/r6/tuning_c64b.json timeout 600 python scripts/slots_sweep.py 5x4 1x4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/sweep_c64b.txt
