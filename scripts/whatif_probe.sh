#!/bin/bash
# What-if upper bounds at the timed operating point (round 6): the frame rate of the 5 x 4 program with one cost REMOVED -- GroupNorm not
# launched, reducers not launched, self-/cross-attention not launched, the unsplit GEMM-form layers leaving their kernels before the
# epilogue.  A probe build of the library (python -m videosd_amd.build --whatif: -DVSD_PROBE, videosd_amd/libvsd_probe.so, never libvsd.so) reads the switches from the
# environment; results are garbage, only the timing is read.  What a perfect fusion / a free epilogue could return, with every
# interaction of the four lanes included.
mkdir -p gpurun_out/r6f
export VSD_LIB=videosd_amd/libvsd_probe.so
for SW in "" VSD_SKIP_GN VSD_SKIP_REDUCE VSD_SKIP_ATTN VSD_SKIP_EPI "VSD_SKIP_GN VSD_SKIP_REDUCE"; do
  echo "== removed: ${SW:-nothing (the probe build as is)}"
  ( for v in $SW; do export $v=1; done; timeout 300 python scripts/slots_sweep.py 5x4 1x4 2>&1 | grep -v amdgpu.ids )
done 2>&1 | tee gpurun_out/r6f/whatif_probe.txt
