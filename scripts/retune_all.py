"""Re-measure the whole per-shape kernel table (profiles/tuning_mi355x.json) on this GPU: after a kernel change the old
choices (tile, split-K, reduction form, pipeline) are the previous kernels' optima.  Starts from an EMPTY table and prepares
every plan the bench, the API leg, the tests and scripts/bench_configs.py use; `Engine.autotune` times every candidate of every
conv shape once.  usage (GPU box): python scripts/retune_all.py [out.json]   (a few minutes)"""
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
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
plans = ([(512, 512, 4, b, True) for b in (5, 1, 2, 3, 4, 8, 6)] + [(512, 512, 4, b, False) for b in (5, 1)] +
         [(768, 768, 8, 1, True), (432, 768, 4, 1, True), (256, 256, 1, 1, True), (192, 256, 2, 1, True), (192, 256, 2, 3, True)])
t_all = time.time()
for (h, w, steps, b, cn) in plans:
    t0 = time.time()
    eng.prepare(h, w, steps, 0.6, use_controlnet=cn, use_graph=False, batch=b)
    print(f"{h}x{w} steps={steps} batch={b} cn={cn}: table {len(ops.tile_override)} entries ({time.time() - t0:.1f} s)", flush=True)
# the coalesced plans again in THROUGHPUT mode (every candidate with four copies in flight on the four launch lanes: the key's last
# field = 1): what bench.py and a worker with three or four lanes run for launches of more than one frame
eng.tune_for_lanes = True
ops.tune_lanes_online = True
for (h, w, steps, b, cn) in [(512, 512, 4, b, True) for b in (5, 2, 3, 4, 8, 6)] + [(512, 512, 4, 5, False)]:
    t0 = time.time()
    eng.prepare(h, w, steps, 0.6, use_controlnet=cn, use_graph=False, batch=b)
    print(f"[four lanes busy] {h}x{w} steps={steps} batch={b} cn={cn}: table {len(ops.tile_override)} entries ({time.time() - t0:.1f} s)", flush=True)
eng.tune_for_lanes = False
del eng
# SDXL-base 1024x1024 (BASELINE configs[3]; scripts/bench_configs.py)
g = torch.Generator().manual_seed(11)
xl = Engine(ops, C.SDXL_UNET, None, C.TAESD, W.synthesize(W.unet_spec(C.SDXL_UNET), "sdxl.", device="cuda"), None, wv)
xl.set_text_embeds((torch.randn(77, 2048, generator=g) * 0.5).half())
xl.set_added_cond((torch.randn(1280, generator=g) * 0.5).half(), (1024, 1024, 0, 0, 1024, 1024))
t0 = time.time()
xl.prepare(1024, 1024, 4, 0.6, use_controlnet=False, use_graph=False)
print(f"SDXL 1024x1024: table {len(ops.tile_override)} entries ({time.time() - t0:.1f} s)", flush=True)
import json  # noqa: E402

json.dump({"device": torch.cuda.get_device_name(ops.device), "table": [[list(k), list(v)] for k, v in sorted(ops.tile_override.items(), key=str)]},
          open(out, "w"), indent=0)
print(f"{len(ops.tile_override)} entries -> {out} ({time.time() - t_all:.0f} s)")
