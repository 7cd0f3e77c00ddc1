#!/bin/bash
mkdir -p gpurun_out/r5
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -3
echo "== stress: first non-finite"; timeout 600 python scripts/find_nonfinite.py --stress 2>&1 | grep -v amdgpu.ids | tail -16 | cut -c1-250
echo "== stress golden"; cp tests/golden/fullsize_oracle.npz gpurun_out/r5/fullsize_oracle.npz; timeout 1500 python scripts/make_fullsize_golden.py gpurun_out/r5/fullsize_oracle.npz --only stress512 2>&1 | grep -v amdgpu.ids | tail -3
cp gpurun_out/r5/fullsize_oracle.npz tests/golden/fullsize_oracle.npz
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -k "range_stress" 2>&1 | tail -12 | cut -c1-300
echo "== one-frame retune"; timeout 900 python scripts/lone_frame.py --tag e1_retune --retune --lanes --save-tuning gpurun_out/r5/tuning_e1.json 2>&1 | tail -1 | cut -c1-900
