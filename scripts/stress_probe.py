"""The range-stress weight set (weights.synthesize(stress=True)) through the engine with parts switched to their unfused forms, against
the oracle's stored full-size output (tests/golden/fullsize_oracle.npz, case stress512): which part of the engine carries the
difference.   python scripts/stress_probe.py [absorb=0] [tail=0] [table=0]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videosd_amd import config as C, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402
from golden_guard import CASES  # noqa: E402
from test_pipeline_gpu import _compare_golden, _frame  # noqa: E402

flags = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
c = CASES["stress512"]
ops = HipOps(0)
if flags.get("table", "1") == "1":
    ops.load_tuning(os.path.join(ROOT, "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda", stress=True)
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda", stress=True)
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.absorb_cross_attention = flags.get("absorb", "1") == "1"
eng.use_fused_tail = flags.get("tail", "1") == "1"
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(c["text_seed"])) * 0.5).half())
eng.prepare(c["H"], c["W"], c["steps"], c["strength"], controlnet_scale=c["cn_scale"], use_controlnet=True, autotune=flags.get("table", "1") == "1")
r0, r1, mad, psnr, got = _compare_golden(eng, _frame(c["H"], c["W"], seed=c["frame_seed"]), c["H"], c["W"], "stress512")
print(f"flags {flags}: init-latent rel-L2 {r0:.2e}, denoised rel-L2 {r1:.3e}, image mean |diff| {mad:.2f} LSB, PSNR {psnr:.1f} dB", flush=True)
