#!/bin/bash
# throughput-mode pairs in their members' own forms: the 5 x 4 headline A/B on ONE box (twice each, alternating), then the workgroup time
mkdir -p gpurun_out/r5
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms")}, d["config"].get("kernel_launches_by_form"))
PY
}
for rep in 1 2; do
  echo "== own-form pairs ($rep)"; timeout 600 python bench.py --no-cpu-baseline --no-api --no-extras > gpurun_out/r5/ab_new$rep.json 2>/dev/null; show gpurun_out/r5/ab_new$rep.json
  echo "== VSD_NO_TWIN ($rep)"; VSD_NO_TWIN=1 timeout 600 python bench.py --no-cpu-baseline --no-api --no-extras > gpurun_out/r5/ab_notwin$rep.json 2>/dev/null; show gpurun_out/r5/ab_notwin$rep.json
  echo "== VSD_NO_GROUP_SHORTCUT ($rep)"; VSD_NO_GROUP_SHORTCUT=1 timeout 600 python bench.py --no-cpu-baseline --no-api --no-extras > gpurun_out/r5/ab_nosc$rep.json 2>/dev/null; show gpurun_out/r5/ab_nosc$rep.json
done
echo "== pair / group op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dropin_gpu.py -x -q -k "group or twin or pair or same_bits" 2>&1 | tail -3
