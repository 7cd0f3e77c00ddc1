#!/bin/bash
# after the grouped kernel's scratch fix and the fused GroupNorm's launch bounds: tests, the table's GROUP entries timed again (the conv entries
# stay), lone frame / 1 x 3 / 1 x 4 before and after the re-tune
mkdir -p gpurun_out/r5
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "group or twin or pair or groupnorm or own" 2>&1 | tail -3
echo "== lone frame, old group entries"; timeout 900 python scripts/lone_frame.py --tag scratchfix_oldtable --lanes 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('p50_ms','gpu_p50_ms','serial_ms','fps_1x3','fps_1x4','launches','launches_one_stream_form')})"
python - <<'PY'
import json
p="profiles/tuning_mi355x.json"
d=json.load(open(p))
n=len(d["table"])
d["table"]=[[k,v] for k,v in d["table"] if k[0]!="group"]
json.dump(d,open(p,"w"),indent=0)
print("group entries dropped:", n-len(d["table"]))
PY
echo "== table"; timeout 1500 python scripts/update_tuning.py 2>&1 | grep -v amdgpu.ids | tail -2; cp profiles/tuning_mi355x.json gpurun_out/r5/tuning_mi355x_regrouped.json
echo "== lone frame, re-timed group entries"; timeout 900 python scripts/lone_frame.py --tag scratchfix_newtable --lanes 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('p50_ms','gpu_p50_ms','serial_ms','fps_1x3','fps_1x4','launches','launches_one_stream_form')})"
echo "== parity"; timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -k "mini or same_bits or baseline_config2_512" 2>&1 | tail -3
