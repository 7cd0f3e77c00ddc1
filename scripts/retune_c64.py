"""Re-time the tuning-table entries of the TAESD block convs (3x3, 64 -> 64 channels) after the persistent form (pipeline 10,
csrc/conv_c64.hip) joined the candidates -- round 6.  Both modes: alone (key's last field 0) and with four lanes busy (1).
Every other entry stays.  usage (GPU box): python scripts/retune_c64.py [out.json]"""
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
path = os.environ.get("VSD_TUNING") or os.path.join(root, "profiles", "tuning_mi355x.json")
out = sys.argv[1] if len(sys.argv) > 1 else path
ops = HipOps(0)
ops.load_tuning(path)
old = dict(ops.tile_override)
is_c64 = lambda k: k[0] != "group" and (k[1] == 64 or k[1] <= 8) and k[2] == 576 and k[3] == 3 and k[4] == 1  # noqa: E731  (M, N, Kp, ksize, stride, ...)
drop = [k for k in ops.tile_override if is_c64(k)]
for k in drop:
    del ops.tile_override[k]
print(f"{len(old)} entries, {len(drop)} entries of 64 -> 64 channel 3x3 convs to re-time", flush=True)
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
budget = float(os.environ.get("VSD_RETUNE_SECONDS", "1e9"))
t_all = time.time()
plans = [(512, 512, 4, b, False) for b in (1, 5, 2, 3, 4, 8, 6)] + [(256, 256, 1, 1, False), (768, 768, 8, 1, False), (432, 768, 4, 1, False)]
plans += [(512, 512, 4, b, True) for b in (5, 2, 3, 4, 8, 6)]
for (h, w, steps, b, lanes) in plans:
    if time.time() - t_all > budget:
        print(f"budget spent before {h}x{w} batch={b} lanes={lanes}: its old entries are kept", flush=True)
        break
    t0 = time.time()
    eng.tune_for_lanes = lanes
    ops.tune_lanes_online = lanes
    eng.prepare(h, w, steps, 0.6, use_controlnet=True, use_graph=False, batch=b)
    print(f"{h}x{w} steps={steps} batch={b} four-lanes={lanes}: table {len(ops.tile_override)} entries ({time.time() - t0:.1f} s)", flush=True)
for k, v in old.items():  # (anything not re-timed keeps its old entry)
    ops.tile_override.setdefault(k, v)
new = [k for k in ops.tile_override if is_c64(k)]
for k in sorted(new, key=str):
    print(f"  M={k[0]:8d} resize={int(k[5])} epi={k[8]} mode={k[9]}: {old.get(k)} -> {ops.tile_override[k]}")
c64 = sum(1 for k in new if ops.tile_override[k][3] == 10)
json.dump({"device": torch.cuda.get_device_name(ops.device), "table": [[list(k), list(v)] for k, v in sorted(ops.tile_override.items(), key=str)]},
          open(out, "w"), indent=0)
print(f"{len(ops.tile_override)} entries -> {out}: {c64} of {len(new)} 64-channel entries now name the persistent form ({time.time() - t_all:.0f} s)")
