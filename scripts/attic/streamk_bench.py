"""Stream-K persistent form (pipeline 8) against the one-tile-per-workgroup forms on the short-K linear layers (and two 3x3
layers) of the 512x512 program: the autotuner's candidate table per shape, best of each family.
usage (GPU box): python scripts/streamk_bench.py [batch=5]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.ops import Geom, HipOps  # noqa: E402
from videosd_amd.packing import pack_conv, pack_linear  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ops = HipOps(0)
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()  # noqa: E731
shapes = [("out-proj 16x16", 256 * B, 1280, 1280, 1), ("out-proj 32x32", 1024 * B, 640, 640, 1), ("proj 64x64", 4096 * B, 320, 320, 1),
          ("qkv 32x32", 1024 * B, 1920, 640, 1), ("qkv 64x64", 4096 * B, 960, 320, 1), ("8x8 linear", 64 * B, 1280, 1280, 1),
          ("xattn out 32x32", 1024 * B, 640, 1024, 1), ("xattn out 16x16", 256 * B, 1280, 1024, 1), ("ff2 16x16", 256 * B, 1280, 5120, 1),
          ("ff2 32x32", 1024 * B, 640, 2560, 1), ("conv3x3 32x32 640", 1024 * B, 640, 640, 3), ("conv3x3 16x16 1280", 256 * B, 1280, 1280, 3)]
for name, m, n, cin, ks in shapes:
    if ks == 1:
        pw = ops.to_device_pack(pack_linear(r(n, cin), r(n)))
        g = Geom.linear(m)
        k = cin
    else:
        hw = m // B
        side = int(hw ** 0.5)
        pw = ops.to_device_pack(pack_conv(r(n, cin, 3, 3), r(n)))
        g = Geom.conv(side, side, batch=B)
        k = cin * 9
    x = r(m, cin).cuda()
    res = r(m, n).cuda()
    out = torch.zeros(m, n, dtype=torch.float16, device="cuda")
    kw = dict(residual=res)
    if ks == 1 and n % 64 == 0:
        kw["rowstat_out"] = torch.zeros(m, n // 64, 2, dtype=torch.float32, device="cuda")
    best, table = ops.tune_conv((x, None, g, pw, out), kw)
    fl = 2.0 * m * n * k
    b8 = next((t for t in table if t[4] == 8), None)
    bo = next((t for t in table if t[4] != 8), None)
    fmt = lambda t: "-" if t is None else f"{t[0]:6.1f} us {fl / t[0] / 1e6:5.0f} TF/s (tile {t[1]} split/parts {t[2]} pipe {t[4]})"  # noqa: E731
    print(f"{name:20s} M={m:6d} N={n:5d} K={k:6d}: tiled {fmt(bo)} | stream-K {fmt(b8)}", flush=True)
    sk = [t for t in table if t[4] == 8][:4]
    print("      stream-K candidates: " + ", ".join(f"{t[0]:.1f} (t{t[1]} p{t[2]})" for t in sk), flush=True)
