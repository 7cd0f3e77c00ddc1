"""Workgroup timeline of ONE layer in the tiled form (pipeline 3) and the persistent stream-K form (pipeline 8): medians of a
workgroup's prologue / K loop / accumulators -> LDS / residual wait / walk + stores, in us.
usage (GPU box): VSD_LIB=videosd_amd/libvsd_tl.so python scripts/sk_timeline.py   (python -m videosd_amd.build --timeline first)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("VSD_LIB"), "set VSD_LIB=videosd_amd/libvsd_tl.so"
from videosd_amd.ops import Geom, HipOps  # noqa: E402
from videosd_amd.packing import pack_linear  # noqa: E402

ops = HipOps(0)
lib = ops.ctx.lib
lib.vsd_wgtl_set.argtypes = [C.c_void_p, C.c_int64]
lib.vsd_wgtl_set.restype = None
lib.vsd_wgtl_used.restype = C.c_int64
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()  # noqa: E731
words = 8 << 20
log = torch.zeros(words, dtype=torch.int64, device="cuda")
for m, n, k, tile in ((320, 1280, 1280, 2), (1280, 1280, 1280, 2), (5120, 1920, 640, 3), (5120, 5120, 640, 3)):
    pw = ops.to_device_pack(pack_linear(r(n, k), r(n)))
    x, res = r(m, k).cuda(), r(m, n).cuda()
    out = torch.zeros(m, n, dtype=torch.float16, device="cuda")
    for pl, sp in ((3, 1), (8, 1)):
        for _ in range(5):
            ops.conv(x, None, Geom.linear(m), pw, out, residual=res, tile=tile, split_k=sp, pipeline=pl)
        ops.synchronize()
        log.zero_()
        torch.cuda.synchronize()
        lib.vsd_wgtl_set(C.c_void_p(log.data_ptr()), words)
        for _ in range(4):
            ops.conv(x, None, Geom.linear(m), pw, out, residual=res, tile=tile, split_k=sp, pipeline=pl)
        ops.synchronize()
        torch.cuda.synchronize()
        used = int(lib.vsd_wgtl_used())
        lib.vsd_wgtl_set(None, 0)
        host = log[:used].cpu().numpy().astype(np.uint64)
        pos, rows = 0, []
        while pos < used:
            grid = int(host[pos])
            rows.append(host[pos + 2:pos + 2 + 8 * grid].reshape(grid, 8).astype(np.int64))
            pos += 2 + 8 * grid
        rr = rows[-1]  # the last of the four launches
        t0, t1, t2, ta, tb, tc = rr[:, 0], rr[:, 1], rr[:, 2], rr[:, 4], rr[:, 5], rr[:, 6]
        ok = t2 > 0
        med = lambda a, b: float(np.median((b[ok] - a[ok]) / 100.0))  # noqa: E731
        span = (t2[ok].max() - t0[ok].min()) / 100.0
        skew = (t0[ok].max() - t0[ok].min()) / 100.0
        print(f"M={m} N={n} K={k} tile={tile} pipeline={pl}: grid {len(rr)} span {span:.1f} us, dispatch skew {skew:.1f}, life {med(t0, t2):.1f} = "
              f"prologue {med(t0, tc):.1f} + loop {med(tc, t1):.1f} + acc->lds {med(t1, ta):.1f} + residual wait {med(ta, tb):.1f} + walk/stores {med(tb, t2):.1f}", flush=True)
