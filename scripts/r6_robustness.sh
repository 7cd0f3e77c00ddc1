#!/bin/bash
# round 6, final build: the fuzzers and soaks (the persistent 64-channel conv among the fuzzers' cases and in every TAESD of the programs)
mkdir -p gpurun_out/r6f
{
echo "== guard_page_fuzz 150 s seed 5"; timeout 400 python scripts/guard_page_fuzz.py 150 5 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600
echo "== guard_page_fuzz 120 s seed 11"; timeout 400 python scripts/guard_page_fuzz.py 120 11 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600
echo "== guard_fuzz 90 s"; timeout 300 python scripts/guard_fuzz.py 90 4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
echo "== guard_page_engine (table, lanes)"; timeout 900 python scripts/guard_page_engine.py table lanes 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
echo "== option_fuzz 120 s, table mode"; timeout 400 python scripts/option_fuzz.py 120 3 table 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
echo "== api_soak 150 s, 4 lanes, up to 5 frames per launch"; timeout 600 python scripts/api_soak.py 150 4 5 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
echo "== soak: four one-frame launches in flight (lock-step form)"; SOAK_BATCH=1 SOAK_LANES=4 timeout 300 python scripts/soak.py 400 2>&1 | grep -v amdgpu.ids | tail -2
echo "== soak: two one-frame launches with the side stream"; SOAK_BATCH=1 SOAK_SIDE=1 timeout 300 python scripts/soak.py 300 2>&1 | grep -v amdgpu.ids | tail -2
echo "== soak: 5 x 4"; SOAK_BATCH=5 SOAK_LANES=4 timeout 300 python scripts/soak.py 80 2>&1 | grep -v amdgpu.ids | tail -2
} 2>&1 | tee gpurun_out/r6f/robustness_runs.txt
