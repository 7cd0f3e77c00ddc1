#!/bin/bash
# round 6: a longer in-situ pass over the throughput-mode table (more shapes, more alternatives), then A/B/A/B of the bench line
mkdir -p gpurun_out/r6i
cp profiles/tuning_mi355x.json /tmp/table_a.json
b() { timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('value_long'))"; }
b A0 | tee gpurun_out/r6i/ab.txt
timeout 3300 python scripts/tune_in_situ.py --shapes 90 --alts 5 --seconds 2700 gpurun_out/r6i/tuning_insitu2.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6i/tune_in_situ2.txt | tail -15
if [ -s gpurun_out/r6i/tuning_insitu2.json ]; then
  for i in 1 2; do
    cp gpurun_out/r6i/tuning_insitu2.json profiles/tuning_mi355x.json; b B$i | tee -a gpurun_out/r6i/ab.txt
    cp /tmp/table_a.json profiles/tuning_mi355x.json; b A$i | tee -a gpurun_out/r6i/ab.txt
  done
fi
