#!/bin/bash
mkdir -p gpurun_out/r5
echo "== new tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv_group or twin or pair or splitk" 2>&1 | tail -8
echo "== ops suite"; timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -4
echo "== lone frame, twin"; timeout 900 python scripts/lone_frame.py --tag twin --lanes --save-tuning gpurun_out/r5/tuning_twin.json 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900
echo "== lone frame, VSD_NO_TWIN"; VSD_NO_TWIN=1 timeout 900 python scripts/lone_frame.py --tag notwin --lanes 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900
echo "== pipeline parity"; timeout 1500 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -4
