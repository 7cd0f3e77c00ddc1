"""Re-time the THROUGHPUT-mode entries of profiles/tuning_mi355x.json (key's last field 1: candidates timed with four launch lanes
busy) after a change of the candidate set -- round 6: the eight-wave forms and the 256 x 256 tile.  The latency-mode entries (field
0) and the group / pair entries stay.  usage (GPU box): python scripts/retune_mode1.py [out.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "tuning_mi355x.json")
out = sys.argv[1] if len(sys.argv) > 1 else path
ops = HipOps(0)
ops.load_tuning(path)
old = dict(ops.tile_override)
drop = [k for k in ops.tile_override if k[0] != "group" and k[-1] == 1]
for k in drop:
    del ops.tile_override[k]
print(f"{len(old)} entries, {len(drop)} throughput-mode conv entries to re-time", flush=True)
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.tune_for_lanes = True
ops.tune_lanes_online = True
budget = float(os.environ.get("VSD_RETUNE_SECONDS", "1e9"))
t_all = time.time()
for (h, w, steps, b, cn) in [(512, 512, 4, b, True) for b in (5, 2, 3, 4, 8, 6)] + [(512, 512, 4, 5, False)]:
    if time.time() - t_all > budget:
        print(f"budget spent before batch={b} cn={cn}: its old entries are kept", flush=True)
        break
    t0 = time.time()
    eng.prepare(h, w, steps, 0.6, use_controlnet=cn, use_graph=False, batch=b)
    print(f"[four lanes busy] {h}x{w} steps={steps} batch={b} cn={cn}: table {len(ops.tile_override)} entries ({time.time() - t0:.1f} s)", flush=True)
    ops.save_tuning(out + ".partial")
for k, v in old.items():  # (anything not re-timed keeps its old entry)
    ops.tile_override.setdefault(k, v)
changed = sum(1 for k in drop if ops.tile_override.get(k) != old[k])
w8 = sum(1 for k in drop if ops.tile_override[k][3] >= 8)
json.dump({"device": torch.cuda.get_device_name(ops.device), "table": [[list(k), list(v)] for k, v in sorted(ops.tile_override.items(), key=str)]},
          open(out, "w"), indent=0)
print(f"{len(ops.tile_override)} entries -> {out}: {changed} of {len(drop)} throughput-mode entries changed, {w8} now name an eight-wave form ({time.time() - t_all:.0f} s)")
