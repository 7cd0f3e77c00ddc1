"""Register / LDS / scratch use of every kernel of libvsd (device-only compile to ISA, no GPU needed):
    python scripts/kernel_resources.py [file.hip ...] [--max-scratch=0]
Prints one line per kernel: VGPRs (arch + accumulator), SGPRs, LDS bytes, scratch bytes per lane, spills.  A kernel with
scratch (private segment) has a local array the compiler could not keep in registers -- every access is a memory round
trip (cdna guide, rule 20); the library is meant to have none.  Exit code 1 when a kernel exceeds --max-scratch."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd.build import CSRC, EXTRA_FLAGS, SOURCES  # noqa: E402

files = [a for a in sys.argv[1:] if a.endswith(".hip")] or SOURCES
max_scratch = None
for a in sys.argv[1:]:
    if a.startswith("--max-scratch="):
        max_scratch = int(a.split("=")[1])
tmp = tempfile.mkdtemp(prefix="vsd_isa_")


def one(f):
    out = os.path.join(tmp, os.path.basename(f).replace(".hip", ".s"))
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-S", "--cuda-device-only", "-w"] + \
        EXTRA_FLAGS.get(os.path.basename(f), []) + ["-o", out, f if os.path.isabs(f) else os.path.join(CSRC, f)]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    rows = []
    cur = {}
    for ln in open(out):
        m = re.match(r"\s+\.(name|vgpr_count|agpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s+(\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "name" and "(" not in v and not v.startswith("'"):
            if v.startswith("_Z") or v.endswith("kernel"):
                cur["name"] = v
        else:
            cur[k] = v
        if k == "vgpr_spill_count":  # last field of a kernel's metadata block
            rows.append(cur)
            cur = {}
    return f, rows


bad = 0
with ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:
    for f, rows in ex.map(one, files):
        print(f"== {os.path.basename(f)}")
        for r in rows:
            name = subprocess.run(["c++filt", r.get("name", "?")], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").replace("(ConvParams)", "")[:110]
            scratch = int(r.get("private_segment_fixed_size", 0))
            flag = "  <-- SCRATCH" if scratch else ""
            if max_scratch is not None and scratch > max_scratch:
                bad += 1
            print(f"  vgpr {int(r.get('vgpr_count', 0)):3d} (acc {int(r.get('agpr_count', 0)):3d}) sgpr {int(r.get('sgpr_count', 0)):3d} lds {int(r.get('group_segment_fixed_size', 0)):6d} "
                  f"scratch {scratch:4d} spill {int(r.get('vgpr_spill_count', 0)):3d}  {name}{flag}")
sys.exit(1 if bad else 0)
