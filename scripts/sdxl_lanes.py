"""SDXL 1024x1024 4-step (BASELINE configs[3]) on 1 / 2 / 4 launch lanes: python scripts/sdxl_lanes.py   (VSD_STREAMS=plain for torch streams)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps
ops = HipOps(0)
ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
wx = W.synthesize(W.unet_spec(C.SDXL_UNET), "sdxl.", device="cuda")
xl = Engine(ops, C.SDXL_UNET, None, C.TAESD, wx, None, wv)
g = torch.Generator().manual_seed(11)
xl.set_text_embeds((torch.randn(77, 2048, generator=g) * 0.5).half())
xl.set_added_cond((torch.randn(1280, generator=g) * 0.5).half(), (1024, 1024, 0, 0, 1024, 1024))
f = np.random.default_rng(0).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
engs = [xl]
for n in (1, 2, 3, 4):
    while len(engs) < n:
        engs.append(xl.make_slot())
    for e in engs[:n]:
        if e.plan is None:
            e.prepare(1024, 1024, 4, 0.6, use_controlnet=False)
            e.ops.upload(e.frame_u8, torch.from_numpy(f))
    for i in range(2 * n): engs[i % n].launch()
    for e in engs[:n]: e.ops.synchronize()
    t = time.perf_counter()
    for i in range(12): engs[i % n].launch()
    for e in engs[:n]: e.ops.synchronize()
    fps = 12 / (time.perf_counter() - t)
    print(f"SDXL 1024x1024 4-step, {n} lane(s): {fps:.2f} frames/s = {27.04 * fps / 2500:.3f} of the MFMA peak", flush=True)
