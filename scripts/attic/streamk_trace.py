"""Kernel durations of the tiled (pipeline 3) and the stream-K (pipeline 8) form on a few shapes, for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/r4/sk_trace -- python3 scripts/streamk_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.ops import Geom, HipOps  # noqa: E402
from videosd_amd.packing import pack_linear  # noqa: E402

ops = HipOps(0)
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()  # noqa: E731
for m, n, k, tile in ((1280, 1280, 1280, 2), (5120, 1920, 640, 3), (320, 1280, 1280, 2)):
    pw = ops.to_device_pack(pack_linear(r(n, k), r(n)))
    x, res = r(m, k).cuda(), r(m, n).cuda()
    out = torch.zeros(m, n, dtype=torch.float16, device="cuda")
    for pl, sp in ((3, 1), (8, 1), (8, 2)):
        for _ in range(40):
            ops.conv(x, None, Geom.linear(m), pw, out, residual=res, tile=tile, split_k=sp, pipeline=pl)
        ops.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ops.stream)
        for _ in range(40):
            ops.conv(x, None, Geom.linear(m), pw, out, residual=res, tile=tile, split_k=sp, pipeline=pl)
        e1.record(ops.stream)
        e1.synchronize()
        print(f"M={m} N={n} K={k} tile={tile} pipeline={pl} parts={sp}: {e0.elapsed_time(e1) / 40 * 1e3:.1f} us per launch (events)", flush=True)
