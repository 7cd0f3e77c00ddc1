#!/bin/bash
mkdir -p gpurun_out/r6
for L in "" videosd_amd/libvsd_nt.so; do
  echo "== library: ${L:-libvsd.so}"
  ( [ -n "$L" ] && export VSD_LIB=$L; timeout 600 python scripts/slots_sweep.py 5x4 1x4 2>&1 | grep -v amdgpu.ids; timeout 300 python scripts/lone_frame.py --tag nt 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('p50_ms','gpu_p50_ms','serial_ms')})" )
done 2>&1 | tee gpurun_out/r6/nt_stores.txt
echo "== parity with non-temporal stores"; VSD_LIB=videosd_amd/libvsd_nt.so timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "mini_pipeline or same_bits" 2>&1 | tail -3
