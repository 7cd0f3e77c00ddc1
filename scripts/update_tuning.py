"""Add the missing per-shape kernel configurations to profiles/tuning_mi355x.json: prepares the 512x512 4-step program
for 1-4 frames per launch (with and without ControlNet), 768x768 8-step and the UI's 768x432, lets `Engine.autotune`
time the candidates of every conv shape that is not in the table yet (GPU box), and writes the merged table back.
usage: python scripts/update_tuning.py [out.json]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "tuning_mi355x.json")
out = sys.argv[1] if len(sys.argv) > 1 else path
ops = HipOps(0)
n0 = ops.load_tuning(path)
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
plans = [(512, 512, 4, b, cn) for b in (1, 2, 3, 4) for cn in (True, False)] + [(768, 768, 8, 1, True), (432, 768, 4, 1, True),
                                                                                   (256, 256, 1, 1, True)]
for (h, w, steps, b, cn) in plans:
    t0 = time.time()
    eng.prepare(h, w, steps, 0.6, use_controlnet=cn, use_graph=False, batch=b)
    print(f"{h}x{w} steps={steps} batch={b} cn={cn}: table {len(ops.tile_override)} entries ({time.time() - t0:.1f} s)", flush=True)
ops.save_tuning(out)
print(f"loaded {n0}, now {len(ops.tile_override)} entries -> {out}")
