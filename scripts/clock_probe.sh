#!/bin/bash
# Shader clock while the 5 x 4 program runs in another process (scripts/clock_probe.cpp), and on the idle chip
mkdir -p gpurun_out/r6
hipcc --offload-arch=gfx950 -O2 scripts/clock_probe.cpp -o /tmp/clock_probe.bin || exit 1
echo "idle chip:"; /tmp/clock_probe.bin 5
( python scripts/slots_sweep.py 5x4 5x4 5x4 5x4 5x4 5x4 > gpurun_out/r6/clock_probe_sweep.txt 2>&1 ) &
PID=$!
sleep 20
echo "while the sweep process runs (engine build, capture, then 5 x 4 replays):"
/tmp/clock_probe.bin 60
wait $PID
grep -v amdgpu.ids gpurun_out/r6/clock_probe_sweep.txt
