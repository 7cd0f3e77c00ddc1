#!/bin/bash
mkdir -p gpurun_out/r6
echo "== parity"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "register_epilogue or all_tiles_splitk or eight_wave" 2>&1 | tail -12
echo "== probe, four lanes"; timeout 1500 python scripts/w8_probe.py --mode1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/regepi_probe_mode1.txt | awk -F'|' '{print $1 "|" $2 "|" $4}' | cut -c1-260
echo "== probe, alone"; timeout 1500 python scripts/w8_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/regepi_probe_mode0.txt | awk -F'|' '{print $1 "|" $4}' | cut -c1-200
