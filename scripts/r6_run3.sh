#!/bin/bash
mkdir -p gpurun_out/r6
echo "== eight-wave parity"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "eight_wave or 256x128 or all_tiles_splitk or geglu or softmax or group" 2>&1 | tail -8
echo "== one forward on stress weights"; timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "one_forward" -s 2>&1 | grep -v amdgpu.ids | tail -8
echo "== w8 probe lanes"; timeout 1500 python scripts/w8_probe.py --mode1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/w8_probe3_mode1.txt
