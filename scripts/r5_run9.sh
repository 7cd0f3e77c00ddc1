#!/bin/bash
mkdir -p gpurun_out/r5
echo "== group tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv_group or splitk" 2>&1 | tail -4
echo "== lone frame, grouped merges"; timeout 900 python scripts/lone_frame.py --tag group 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300
echo "== lone frame, VSD_NO_GROUP"; VSD_NO_GROUP=1 timeout 900 python scripts/lone_frame.py --tag nogroup 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300
echo "== pipeline parity"; timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -4
