"""Frames per launch x launches in flight: throughput of the 512x512 4-step ControlNet program at several operating points,
one engine build.  usage: python scripts/slots_sweep.py 5x2 5x3 4x3 6x2 8x2   (overlap of the two encoders: on below 3 slots,
as bench.py does; append 'o' / 'n' to force it on / off: 5x3o)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402

pts = [a for a in sys.argv[1:] if "x" in a] or ["5x2", "5x3"]
ops = HipOps(0)
ops.load_tuning(os.environ.get("VSD_TUNING") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
slots = [eng]
for pt in pts:
    force = pt[-1] if pt[-1] in "on" else None
    b, s = (int(x) for x in pt.rstrip("on").split("x"))
    eng.overlap_controlnet = True
    ovl = (s < 3) if force is None else (force == "o")
    while len(slots) < s:
        slots.append(eng.make_slot())
    pool = slots[:s]
    for e in pool:
        e.overlap_controlnet = True
        e.overlap_launch = ovl
        e.tune_for_lanes = s >= 3 and b > 1 and not os.environ.get("VSD_NO_LANE_TUNING")  # (bench.py's and the drop-in class's rule)
        if os.environ.get("SWEEP_LANE_TUNING") is not None:  # (experiments: force the kernel-choice mode)
            e.tune_for_lanes = os.environ["SWEEP_LANE_TUNING"] == "1"
        e.prepare(512, 512, 4, 0.6, use_controlnet=True, batch=b)
    f = np.random.default_rng(0).integers(0, 256, (512, 512, 3) if b == 1 else (b, 512, 512, 3), dtype=np.uint8)
    for e in pool:
        e.infer_u8(f)
    res = []
    for _rep in range(3):
        n = max(4 * s, 60 // b)
        for e in pool:
            e.ops.synchronize()
        t = time.perf_counter()
        for i in range(n):
            pool[i % s].launch()
        for e in pool:
            e.ops.synchronize()
        res.append(n * b / (time.perf_counter() - t))
    where = ""
    if os.environ.get("SWEEP_ADDR"):  # (where the slots' arenas landed: throughput turned out to depend on it)
        where = "  arenas " + " | ".join(",".join(hex(c.data_ptr()) for c in e.arena.chunks[:3]) for e in pool)
    print(f"{pt}: overlap={ovl} fps {max(res):.1f} (runs {', '.join('%.1f' % r for r in res)}){where}", flush=True)
