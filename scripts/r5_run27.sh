#!/bin/bash
# the lock-step pairs at five frames per launch again, now that the grouped kernel's 128-row forms no longer run from scratch memory
mkdir -p gpurun_out/r5
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","value_long","p50_latency_ms")}, d["config"].get("kernel_launches_by_form"))
PY
}
for rep in 1 2; do
  echo "== default ($rep)"; timeout 600 python bench.py --no-cpu-baseline --no-api --no-extras > gpurun_out/r5/ab2_def$rep.json 2>/dev/null; show gpurun_out/r5/ab2_def$rep.json
  echo "== pairs at every batch size, members' own form ($rep)"; VSD_TWIN_ALL=1 VSD_PAIR_OWN_FORM=1 timeout 600 python bench.py --no-cpu-baseline --no-api --no-extras > gpurun_out/r5/ab2_own$rep.json 2>/dev/null; show gpurun_out/r5/ab2_own$rep.json
  echo "== pairs at every batch size, latency-timed forms ($rep)"; VSD_TWIN_ALL=1 timeout 600 python bench.py --no-cpu-baseline --no-api --no-extras > gpurun_out/r5/ab2_lat$rep.json 2>/dev/null; show gpurun_out/r5/ab2_lat$rep.json
done
