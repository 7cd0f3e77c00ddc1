"""Which op of the one-frame program first writes a non-finite value -- and how close to the fp16 range the others come.

    python scripts/find_nonfinite.py [--stress] [--size 512] [--steps 4]

Runs the recorded program of a 512x512 4-step ControlNet frame op by op (eager), after every op looks at what it wrote: the first
output with an Inf / NaN is named with its shape and epilogue, and the ten outputs with the largest |x| are listed (fp16 tops out at
65 504).  --stress: the range-stress weight set (weights.synthesize(stress=True))."""
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
from test_pipeline_gpu import _frame  # noqa: E402

argv = sys.argv[1:]
stress = "--stress" in argv
size = int(argv[argv.index("--size") + 1]) if "--size" in argv else 512
steps = int(argv[argv.index("--steps") + 1]) if "--steps" in argv else 4
ops = HipOps(0)
ops.load_tuning(os.path.join(ROOT, "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda", stress=stress)
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda", stress=stress)
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.overlap_controlnet = False
eng.prepare(size, size, steps, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=False, autotune=False)
ops.upload(eng.frame_u8, torch.from_numpy(_frame(size, size, seed=71)))
eng._sync_prompt()
OUT = {"conv": lambda a, k: [a[4], k.get("out_t"), k.get("out2")], "conv_group": lambda a, k: [m[0][4] for m in a[0]], "groupnorm": lambda a, k: [a[10]], "attention": lambda a, k: [a[6]],
       "tail_a": lambda a, k: [a[5], a[6]], "tail_b": lambda a, k: [a[8]], "lcm_step_dev": lambda a, k: [a[6], a[7]], "add_noise_dev": lambda a, k: [a[5]]}
seen = []
first = None
for i, (fn, a, k) in enumerate(eng.program.calls):
    name = fn.__name__
    if name in Engine.SYNC_OPS:
        continue
    fn(*a, **k)
    if name not in OUT:
        continue
    ops.synchronize()
    for t in OUT[name](a, k):
        if t is None or not t.dtype.is_floating_point:
            continue
        x = t.float()
        fin = bool(torch.isfinite(x).all())
        mx = float(x[torch.isfinite(x)].abs().max()) if bool(torch.isfinite(x).any()) else float("nan")
        desc = f"#{i} {name} out {tuple(t.shape)}"
        if name == "conv":
            g, w = a[2], a[3]
            desc += f" M={g.m} N={w.n} K={w.k} ks={g.ksize} geglu={w.geglu} act={k.get('act', 0)} residual={'y' if k.get('residual') is not None else 'n'} ln={'y' if k.get('ln_part') is not None else 'n'}"
        seen.append((mx, desc))
        if not fin and first is None:
            first = desc
            print("FIRST NON-FINITE:", desc, "| finite max |x| before:", mx, flush=True)
    if first is not None:
        break
print("largest finite |x| written:")
for mx, d in sorted(seen, key=lambda t: -(t[0] if t[0] == t[0] else 0))[:12]:
    print(f"  {mx:12.1f}  {d}")
print("first non-finite:", first)
