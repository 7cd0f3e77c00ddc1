#!/bin/bash
# shortcut convs in conv1's grid (own-split groups) + one-FMA scale/residual: op tests, pipeline parity, lone frame with / without
mkdir -p gpurun_out/r5
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "group or twin or pair or splitk or epilogue or own" 2>&1 | tail -6
echo "== lone frame, shortcuts grouped"; timeout 900 python scripts/lone_frame.py --tag sc_grouped --lanes --save-tuning gpurun_out/r5/tuning_groups.json 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1500
echo "== lone frame, shortcuts alone"; VSD_NO_GROUP_SHORTCUT=1 timeout 900 python scripts/lone_frame.py --tag sc_alone --lanes 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-700
echo "== pipeline parity"; timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_dropin_gpu.py -x -q 2>&1 | tail -4
