"""Print the autotuner's candidate table for the heaviest conv shapes of the 512x512 program."""
import sys, os, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps
ops = HipOps(0)
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.prepare(512, 512, 4, 0.6, use_controlnet=True, use_graph=False, autotune=False)
cnt = collections.Counter(); first = {}
for fn, a, k in Engine.flat_calls(eng.program.calls):
    if fn.__name__ != "conv": continue
    key = ops.conv_key_of(a[2], a[3], k)
    cnt[key] += 1; first.setdefault(key, (a, k))
rows = []
for key, (a, k) in first.items():
    best, table = ops.tune_conv(a, k)
    g, w = a[2], a[3]
    fl = 2.0 * g.m * w.n * w.k
    rows.append((best[0] * cnt[key], key, cnt[key], fl, table))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"sum of best times over the frame: {tot/1e3:.2f} ms")
for t, key, n, fl, table in rows[:14]:
    print(f"\n{key} x{n}: total {t/1e3:.2f} ms; {fl/1e9:.2f} GFLOP each")
    for us, tile, sp, ink, pl in table[:6]:
        print(f"    {us:7.1f} us  {fl/us/1e6:7.1f} TF/s  tile={tile} split={sp} inkernel={ink} pipe={pl}")
    worst = table[-1]
    print(f"    ... worst {worst[0]:.1f} us ({len(table)} candidates)")
