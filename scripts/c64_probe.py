"""The persistent 64-channel conv (pipeline 10) against the halo-patch and GEMM-form candidates on TAESD's layer shapes:
    python scripts/c64_probe.py [--mode1]
For each (frames, image side, Cout, residual + ReLU) the tuner's candidate table: best form per pipeline family."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import lib as L  # noqa: E402
from videosd_amd.ops import Geom, HipOps  # noqa: E402
from videosd_amd.packing import pack_conv  # noqa: E402

ops = HipOps(0)
ops.tune_mode = 1 if "--mode1" in sys.argv else 0
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()  # noqa: E731
for B, side, cout, block_end, up in [(5, 512, 64, False, False), (5, 512, 64, True, False), (5, 256, 64, False, False), (5, 256, 64, False, True), (1, 512, 64, False, False),
                                     (1, 256, 64, True, False), (5, 128, 64, False, False), (5, 512, 3, False, False), (1, 512, 3, False, False)]:
    pw = ops.to_device_pack(pack_conv(r(cout, 64, 3, 3), r(cout)))
    hs = side // 2 if up else side
    g = Geom.conv(hs, hs, batch=B, up_to=(side, side) if up else None)
    x = r(B * hs * hs, 64).cuda()
    ldo = 64 if cout == 64 else 8
    out = torch.zeros(g.m, ldo, dtype=torch.float16, device="cuda")
    kw = dict(act=L.ACT_RELU) if cout == 64 else dict(ldo=8)
    if block_end:
        kw = dict(residual=r(g.m, 64).cuda(), act=L.ACT_RELU | L.ACT_POST)
    best, table = ops.tune_conv((x, None, g, pw, out), kw)
    fl = 2.0 * g.m * cout * 576
    by = {}
    for t in table:
        fam = "persistent" if t[4] == 10 else ("halo" if t[4] == 7 else "gemm")
        by.setdefault(fam, t)
    f = lambda t: "none" if t is None else f"{t[0]:7.1f} us {fl / t[0] / 1e6:6.0f} TF/s {2 * g.m * (64 + ldo) / t[0] / 1e3:6.0f} GB/s (tile {t[1]} pipe {t[4]})"  # noqa: E731
    print(f"B={B} {side}x{side}{' up' if up else ''} Cout={cout}{' +res,ReLU' if block_end else ''} mode{ops.tune_mode}: persistent {f(by.get('persistent'))} | halo {f(by.get('halo'))} | gemm {f(by.get('gemm'))}", flush=True)
    del pw, x, out
