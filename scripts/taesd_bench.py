"""The 64 -> 64 channel 3x3 convs of TAESD: weights-resident persistent form (pipeline 10) against the tuned halo / GEMM
forms, back-to-back launches, HIP events around 20 of them."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd.ops import HipOps, Geom
from videosd_amd.packing import pack_conv

ops = HipOps(0)
ops.load_tuning(os.path.join(ROOT, "profiles", "tuning_mi355x.json"))
g_ = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()
pw = ops.to_device_pack(pack_conv(rnd(64, 64, 3, 3), rnd(64)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B, h, w, up in [(5, 512, 512, None), (5, 256, 256, None), (5, 128, 128, None), (5, 64, 64, None), (1, 512, 512, None), (1, 256, 256, None),
                    (5, 256, 256, (512, 512)), (3, 512, 512, None)]:
    g = Geom.conv(h, w, up_to=up, batch=B)
    x = rnd(B * h * w, 64).cuda(); res = rnd(g.m, 64).cuda()
    row = []
    outs = []
    for name, kw in [("table", {}), ("halo 128x64", {"pipeline": 7, "tile": 1, "split_k": 1}), ("resident", {"pipeline": 10, "tile": 1})]:
        out = torch.zeros(g.m, 64, dtype=torch.float16, device="cuda")
        f = lambda: ops.conv(x, None, g, pw, out, act=1 | 256, residual=res, **kw)
        for _ in range(3): f()
        ops.synchronize()
        best = 1e9
        for _ in range(3):
            e0.record(ops.stream)
            for _ in range(20): f()
            e1.record(ops.stream); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        outs.append(out)
        row.append(f"{name}: {best:7.1f} us ({2.0 * g.m * 64 * 576 / best / 1e6:5.0f} TF/s)")
    same = torch.equal(outs[1], outs[2])
    print(f"B={B} {h}x{w} up={up} M={g.m} | " + " | ".join(row) + f" | resident == halo: {same}", flush=True)
