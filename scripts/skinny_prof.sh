#!/bin/bash
# per-kernel durations of scripts/skinny_bench.py (gpurun box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/skp -- python3 scripts/skinny_bench.py > gpurun_out/skinny_prof.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/skp/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
