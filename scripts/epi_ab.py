"""A/B of one library build against another on the layer shapes of the 5-frame program (same process order, same box):
    VSD_LIB=videosd_amd/libvsd_oldepi.so python scripts/epi_ab.py   vs   python scripts/epi_ab.py
prints us per launch of fixed (tile, pipeline) choices -- the shipped table's -- for a few heavy conv layers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import lib as L  # noqa: E402
from videosd_amd.ops import Geom, HipOps  # noqa: E402
from videosd_amd.packing import pack_conv, pack_geglu, pack_linear  # noqa: E402

ops = HipOps(0)
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()  # noqa: E731
B = 5


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ops.stream)
        for _ in range(reps):
            fn()
        e1.record(ops.stream)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


rows = []
# GEGLU 5120 x 5120 x 640 (tile 64x128, pipeline 3)
m, c = 1024 * B, 640
pw = ops.to_device_pack(pack_geglu(r(8 * c, c), r(8 * c)))
x = r(m, c).cuda()
out = torch.zeros(m, 4 * c, dtype=torch.float16, device="cuda")
rows.append(("geglu 5120x5120x640 t3 p3", timeit(lambda: ops.conv(x, None, Geom.linear(m), pw, out, tile=3, pipeline=3))))
# 3x3 32x32 640->640 (tile 64x128, pipeline 3), rowvec + SiLU-free plain
pc = ops.to_device_pack(pack_conv(r(640, 640, 3, 3), r(640)))
g = Geom.conv(32, 32, batch=B)
x2 = r(g.m, 640).cuda()
o2 = torch.zeros(g.m, 640, dtype=torch.float16, device="cuda")
rv = r(640).cuda()
rows.append(("conv3x3 5120x640x5760 t3 p3", timeit(lambda: ops.conv(x2, None, g, pc, o2, rowvec=rv, tile=3, pipeline=3))))
# out-proj 1280x1280x1280 + residual + rowstat (tile 64x64 p3), qkv-like 5120x1920x640 (t3 p3), ff2 1280x1280x5120 (t2 p3)
for name, m, n, k, tile in (("out-proj 1280x1280x1280 t2 p3", 1280, 1280, 1280, 2), ("linear 5120x1920x640 t3 p3", 5120, 1920, 640, 3),
                            ("ff2 1280x1280x5120 t2 p3", 1280, 1280, 5120, 2), ("linear 20480x320x320 t3 p3", 20480, 320, 320, 3)):
    pl = ops.to_device_pack(pack_linear(r(n, k), r(n)))
    xx, res = r(m, k).cuda(), r(m, n).cuda()
    oo = torch.zeros(m, n, dtype=torch.float16, device="cuda")
    rs = torch.zeros(m, n // 64, 2, dtype=torch.float32, device="cuda")
    rows.append((name, timeit(lambda: ops.conv(xx, None, Geom.linear(m), pl, oo, residual=res, rowstat_out=rs, tile=tile, pipeline=3))))
# 256x128 tile: 3x3 64x64 320->320 via GEMM form p3
pc2 = ops.to_device_pack(pack_conv(r(320, 320, 3, 3), r(320)))
g2 = Geom.conv(64, 64, batch=B)
x3 = r(g2.m, 320).cuda()
o3 = torch.zeros(g2.m, 320, dtype=torch.float16, device="cuda")
rows.append(("conv3x3 20480x320x2880 t4 p3", timeit(lambda: ops.conv(x3, None, g2, pc2, o3, rowvec=r(320).cuda(), tile=4, pipeline=3))))
rows.append(("conv3x3 20480x320x2880 t0 p3", timeit(lambda: ops.conv(x3, None, g2, pc2, o3, rowvec=r(320).cuda(), tile=0, pipeline=3))))
print(os.environ.get("VSD_LIB", "libvsd.so"))
for n, t in rows:
    print(f"  {n:34s} {t:7.1f} us")
