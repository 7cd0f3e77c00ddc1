"""Development probe: every own-split group (a ResnetBlock's conv1 + shortcut conv) of the mini engine's program, run as a group and as
launches of its own on the same inputs -- which group differs, in which member, with which forms."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import config as C, lib as L, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402

ops = HipOps(0)
wu = W.synthesize(W.unet_spec(C.MINI_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.MINI_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, C.MINI_UNET.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.prepare(128, 128, 1, 0.6, use_controlnet=False, use_graph=False, autotune="--tune" in sys.argv)
bad = 0
for fn, a, k in eng.program.calls:
    if fn.__name__ == "conv_group" and k.get("split") == "own":
        fn(*a, **k)
        ops.synchronize()
        got = [m[0][4].clone() for m in a[0]]
        for (aa, kk) in a[0]:
            ops.conv(*aa, **kk)
        ops.synchronize()
        ent = ops.tile_override.get(ops.group_key(a[0], "own"))
        for i, ((aa, kk), g) in enumerate(zip(a[0], got)):
            d = ops.conv(*aa, _desc_only=True, **kk)
            same = torch.equal(aa[4], g)
            err = (aa[4].float() - g.float()).abs().max().item()
            print(f"group entry {ent} member {i}: M={aa[2].m} N={aa[3].n} K={aa[3].k} ks={aa[2].ksize} own form (tile {d.tile}, split {d.split_k}, "
                  f"pipeline {d.pipeline}) {'same bits' if same else f'DIFFERS max abs {err:.4g}'}", flush=True)
            bad += not same
    else:
        fn(*a, **k)
ops.synchronize()
print("groups with a differing member:", bad)
