"""Eight-wave conv forms (pipelines 8 / 9) against the four-wave forms on the layer shapes of the 5-frame and 1-frame programs:
    python scripts/w8_probe.py [--mode1] [--quick]
For each shape: the tuner's candidate table (alone, or with four copies in flight: --mode1), best four-wave form | best eight-wave
form.  Shapes: (M, Cin, Cout, ksize) with M = B * H * W."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.ops import Geom, HipOps  # noqa: E402
from videosd_amd.packing import pack_conv  # noqa: E402

SHAPES = [  # B, H, W, cin, cout, ks
    (5, 32, 32, 640, 640, 3), (5, 64, 64, 320, 320, 3), (5, 16, 16, 1280, 1280, 3), (5, 32, 32, 640, 5120, 1), (5, 16, 16, 1280, 10240, 1),
    (5, 64, 64, 320, 960, 1), (5, 16, 16, 5120, 1280, 1), (5, 32, 32, 640, 640, 1), (5, 32, 32, 640, 1920, 1), (5, 16, 16, 1280, 3840, 1),
    (5, 32, 32, 2560, 640, 1), (5, 16, 16, 1280, 1280, 1), (5, 64, 64, 640, 320, 3), (5, 64, 64, 320, 320, 1), (5, 32, 32, 1280, 640, 3),
    (1, 64, 64, 320, 320, 3), (1, 32, 32, 640, 640, 3), (1, 32, 32, 640, 5120, 1), (1, 64, 64, 320, 960, 1),
]
if "--quick" in sys.argv:
    SHAPES = SHAPES[:6]
ops = HipOps(0)
ops.tune_mode = 1 if "--mode1" in sys.argv else 0
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()  # noqa: E731
for B, H, W, cin, cout, ks in SHAPES:
    pw = ops.to_device_pack(pack_conv(r(cout, cin, ks, ks), r(cout)))
    g = Geom.conv(H, W, ksize=ks, batch=B)
    x = r(g.m, cin).cuda()
    out = torch.zeros(g.m, cout, dtype=torch.float16, device="cuda")
    rv = r(cout).cuda()
    best, table = ops.tune_conv((x, None, g, pw, out), dict(rowvec=rv))
    fl = 2.0 * g.m * cout * cin * ks * ks
    four = next((t for t in table if t[4] < 8), None)
    eight = next((t for t in table if t[4] in (8, 9) and t[1] != 6), None)
    x2 = next((t for t in table if t[4] in (11, 12)), None)
    huge = next((t for t in table if t[1] == 6), None)
    f = lambda t: "none" if t is None else f"{t[0]:7.1f} us {fl / t[0] / 1e6:5.0f} TF/s tile={t[1]} split={t[2]} ink={int(t[3])} pipe={t[4]}"  # noqa: E731
    print(f"M={g.m:6d} N={cout:5d} K={cin * ks * ks:5d} k{ks} mode{ops.tune_mode}: four {f(four)} | eight {f(eight)}"
          f"  ({eight[0] / four[0]:.2f}) | 256x256 {f(huge)}" + (f" ({huge[0] / four[0]:.2f})" if huge else "") +
          f" | two slots, two workgroups per CU {f(x2)}" + (f" ({x2[0] / four[0]:.2f})" if x2 else ""), flush=True)
    del pw, x, out
