"""Every tuner candidate of the one-frame program's small-M 1x1 GEMMs ("put >= 240 CUs on the GEMMs that run on
8-80 workgroups: in-launch split-K 2-4"): us per launch back to back on an idle GPU, best pipeline per (tile, split, reduction form).

    python scripts/small_gemm_candidates.py > profiles/round5_small_gemm_candidates.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import config as C, lib as L, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402

ops = HipOps(0)
ops.load_tuning(os.path.join(ROOT, "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.prepare(512, 512, 4, 0.6, use_controlnet=True, use_graph=False, batch=1)
EPI = ["plain", "ln", "rowstat", "ln+rowstat", "softmax", "general", "plain+act"]
want = {(256, 1280, 1280), (64, 1280, 1280), (256, 1024, 1280), (64, 1024, 1280), (256, 3840, 1280), (64, 1280, 2560), (256, 1280, 5120),
        (256, 1280, 1024), (64, 1280, 1024), (1024, 640, 640), (1024, 640, 1024), (4096, 320, 320)}
seen = set()
print("one-frame program (512x512, 4 steps, ControlNet), M <= 256 1x1 GEMMs: us per launch, alone, back to back (12 launches, best of 2)")
for fn, a, k in Engine.flat_calls(eng.program.calls):
    if fn.__name__ != "conv":
        continue
    g, w = a[2], a[3]
    key = ops.conv_key_of(g, w, k)
    if g.ksize != 1 or (g.m, w.n, w.kp) not in want or key in seen:
        continue
    seen.add(key)
    count = sum(1 for f2, a2, k2 in Engine.flat_calls(eng.program.calls) if f2.__name__ == "conv" and ops.conv_key_of(a2[2], a2[3], k2) == key)
    chosen = ops.tile_override.get(key)
    best, table = ops.tune_conv(a, k)
    rows = {}
    for us, t, sp, ink, pl in table:
        kk = (t, sp, ink if pl != 8 else "lean")
        if kk not in rows or us < rows[kk][0]:
            rows[kk] = (us, pl)
    print(f"\nM={g.m} N={w.n} K={w.k} epilogue={EPI[key[-2]]} x{count} per frame; table choice {chosen}; this run's best "
          f"(tile {best[1]}, split {best[2]}, {'in-launch' if best[3] else 'reducer'}, pipeline {best[4]}) {best[0]:.1f} us")
    for (t, sp, ink), (us, pl) in sorted(rows.items(), key=lambda kv: kv[1][0])[:14]:
        bm, bn = L.TILE_DIMS[t]
        wgs = -(-g.m // bm) * -(-w.n // bn) * (1 if ink == "lean" else sp)  # (the lean form's parts of K are the waves of ONE workgroup)
        how = "in-WG    " if ink == "lean" else ("in-launch" if ink else ("reducer  " if sp > 1 else "-        "))
        print(f"   {us:7.1f} us  tile {bm}x{bn} split {sp:2d} {how} pipeline {pl}  ({wgs} workgroups)")
