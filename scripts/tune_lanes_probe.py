"""Does the best (tile, split, pipeline) of a layer change when four lanes run at once?  The tuner times a candidate ALONE on an idle
GPU (launch latency); with four launches in flight the chip is shared and what counts is the CU-time a candidate costs.  For a few
layer shapes: every candidate alone (us per launch) and four copies at once on the four launch lanes (us per launch-quadruple / 4),
each as captured graphs of 12 launches.     usage (GPU box): python scripts/tune_lanes_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import lib as L  # noqa: E402
from videosd_amd.ops import Geom, HipOps  # noqa: E402
from videosd_amd.packing import pack_conv, pack_linear  # noqa: E402

ops = HipOps(0)
lanes = [ops] + [ops.clone() for _ in range(3)]
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()  # noqa: E731
REPS = 12


def graphs_time(fn_per_lane, n_lanes):
    gs = []
    for o, fn in list(zip(lanes, fn_per_lane))[:n_lanes]:
        fn()
        o.synchronize()
        o.graph_begin()
        for _ in range(REPS):
            fn()
        gs.append((o, o.graph_end()))
    best = 1e9
    for _ in range(3):
        for o, g in gs:
            o.graph_launch(g)
        for o, g in gs:
            o.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            for o, g in gs:
                o.graph_launch(g)
        for o, g in gs:
            o.synchronize()
        best = min(best, (time.perf_counter() - t) / 3 / REPS * 1e6)
    for o, g in gs:
        o.graph_destroy(g)
    return best


shapes = [("out-proj 16x16", 1280, 1280, 1280, 1), ("out-proj 32x32", 5120, 640, 640, 1), ("qkv-like 32x32", 5120, 1920, 640, 1), ("ff2 32x32", 5120, 640, 2560, 1),
          ("ff2 16x16", 1280, 1280, 5120, 1), ("proj 64x64", 20480, 320, 320, 1), ("conv3x3 32x32", 5120, 640, 640, 3), ("conv3x3 16x16", 1280, 1280, 1280, 3)]
B = 5
for name, m, n, cin, ks in shapes:
    if ks == 1:
        pw = ops.to_device_pack(pack_linear(r(n, cin), r(n)))
        g = Geom.linear(m)
    else:
        side = int((m // B) ** 0.5)
        pw = ops.to_device_pack(pack_conv(r(n, cin, 3, 3), r(n)))
        g = Geom.conv(side, side, batch=B)
    xs = [r(m, cin).cuda() for _ in lanes]
    res = r(m, n).cuda()
    outs = [torch.zeros(m, n, dtype=torch.float16, device="cuda") for _ in lanes]
    cands = []
    for t in (L.TILE_128x128, L.TILE_128x64, L.TILE_64x128, L.TILE_64x64, L.TILE_256x128):
        for pl in ((3, 5) if t == L.TILE_256x128 else (3, 4)):
            cands.append((t, 1, pl))
        if ks == 3 and t in (L.TILE_128x128, L.TILE_128x64, L.TILE_256x128):
            cands.append((t, 1, 7))
    rows = []
    for t, sp, pl in cands:
        try:
            fns = [(lambda o=o, x=x, out=out: o.conv(x, None, g, pw, out, residual=res, tile=t, split_k=sp, pipeline=pl)) for o, x, out in zip(lanes, xs, outs)]
            alone = graphs_time(fns, 1)
            four = graphs_time(fns, 4) / 4
            rows.append((alone, four, t, sp, pl))
        except RuntimeError:
            continue
    rows.sort()
    best_alone = rows[0]
    best_four = min(rows, key=lambda x: x[1])
    print(f"{name:16s} M={m} N={n} K={cin * ks * ks}: best alone tile {best_alone[2]} pipe {best_alone[4]}: {best_alone[0]:.1f} us alone, {best_alone[1]:.1f} us/launch on 4 lanes | "
          f"best on 4 lanes tile {best_four[2]} pipe {best_four[4]}: {best_four[0]:.1f} alone, {best_four[1]:.1f} on 4 lanes", flush=True)
    print("      " + "  ".join(f"t{t}p{pl}:{a:.1f}/{f:.1f}" for a, f, t, sp, pl in rows), flush=True)
