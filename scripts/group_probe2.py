"""Development probe: the mini engine's frame with the shortcut convs grouped / alone, eager / captured."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import config as C, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402

wu = W.synthesize(W.unet_spec(C.MINI_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.MINI_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
text = (torch.randn(77, C.MINI_UNET.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5).half()
frame = np.random.default_rng(0).integers(0, 256, (128, 128, 3), dtype=np.uint8)
outs = {}
for cn in (False, True):
    for group in (False, True):
        for graph in (False, True):
            for tune in (False, True):
                ops = HipOps(0)
                eng = Engine(ops, C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
                eng.group_shortcuts = group
                eng.set_text_embeds(text)
                eng.prepare(128, 128, 2, 0.6, use_controlnet=cn, use_graph=graph, autotune=tune)
                o = eng.infer_u8(frame).astype(int)
                den = eng.buffers["denoised"][:, :4].float().cpu()
                outs[(cn, group, graph, tune)] = (o, den)
                ref = outs[(cn, False, False, False)]
                print(f"cn={cn} group={group} graph={graph} tune={tune}: image mean |diff| vs plain {np.abs(o - ref[0]).mean():.3f}, "
                      f"latent rel {float((den - ref[1]).norm() / ref[1].norm()):.4g}", flush=True)
