"""Add the missing per-shape kernel configurations to profiles/tuning_mi355x.json: prepares every plan scripts/retune_all.py prepares
(512x512 4-step at 1-8 frames per launch in both tuning modes, 768x768 8-step, the UI's 768x432, the small test sizes), lets
`Engine.autotune` time the candidates of every conv shape / group / pair that is not in the table yet (GPU box), and writes the merged
table back.
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
# (the plan list of scripts/retune_all.py: what the bench, the API leg, the tests and scripts/bench_configs.py prepare)
plans = ([(512, 512, 4, b, True) for b in (5, 1, 2, 3, 4, 8, 6)] + [(512, 512, 4, b, False) for b in (5, 1)] +
         [(768, 768, 8, 1, True), (432, 768, 4, 1, True), (256, 256, 1, 1, True), (192, 256, 2, 1, True), (192, 256, 2, 3, True)])
for (h, w, steps, b, cn) in plans:
    t0 = time.time()
    eng.prepare(h, w, steps, 0.6, use_controlnet=cn, use_graph=False, batch=b)
    print(f"{h}x{w} steps={steps} batch={b} cn={cn}: table {len(ops.tile_override)} entries ({time.time() - t0:.1f} s)", flush=True)
# ... and the coalesced plans in throughput mode (key's last field 1: candidates timed with four lanes busy; groups / pairs are timed
# alone in either mode) -- only what the table lacks, e.g. the group entries of the lock-step encoders and the shortcut groups
eng.tune_for_lanes = True
ops.tune_lanes_online = True
for (h, w, steps, b, cn) in [(512, 512, 4, b, True) for b in (5, 2, 3, 4, 8, 6)] + [(512, 512, 4, 5, False)]:
    t0 = time.time()
    eng.prepare(h, w, steps, 0.6, use_controlnet=cn, use_graph=False, batch=b)
    print(f"[four lanes busy] {h}x{w} steps={steps} batch={b} cn={cn}: table {len(ops.tile_override)} entries ({time.time() - t0:.1f} s)", flush=True)
eng.tune_for_lanes = False
ops.save_tuning(out)
print(f"loaded {n0}, now {len(ops.tile_override)} entries -> {out}")
