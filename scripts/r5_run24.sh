#!/bin/bash
# longer stability runs on the final build: the drop-in class under session changes, launches in flight against sequential results
mkdir -p gpurun_out/r5
( echo "== api_soak 240 s, 4 lanes, up to 5 frames per launch"; timeout 600 python scripts/api_soak.py 240 4 5 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== api_soak 120 s, memory_budget 0.12"; timeout 400 python scripts/api_soak.py 120 4 5 0.12 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== soak: four one-frame launches in flight (lock-step form) against sequential results"; SOAK_BATCH=1 SOAK_LANES=4 timeout 400 python scripts/soak.py 400 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== soak: two one-frame launches, each with its ControlNet encoder (and its shortcut groups) on the side stream"; SOAK_BATCH=1 SOAK_SIDE=1 timeout 400 python scripts/soak.py 300 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== soak: 5 x 4"; SOAK_BATCH=5 SOAK_LANES=4 timeout 400 python scripts/soak.py 80 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee gpurun_out/r5/soak_runs.txt | cut -c1-400
