"""The small-image layers (M <= 192) on the weight-streaming form (pipeline 9) against the tuning table's choice:
back-to-back launches on one stream, HIP events around 30 of them."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd.ops import HipOps, Geom
from videosd_amd.packing import pack_conv

ops = HipOps(0)
ops.load_tuning(os.path.join(ROOT, "profiles", "tuning_mi355x.json"))
g_ = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()
shapes = [(3, 8, 8, 1280, 0, 1280, 3), (3, 8, 8, 1280, 1280, 1280, 3), (1, 8, 8, 1280, 0, 1280, 3), (1, 8, 8, 1280, 1280, 1280, 3),
          (3, 8, 8, 1280, 0, 1280, 1), (3, 8, 8, 5120, 0, 1280, 1), (3, 8, 8, 1280, 0, 3840, 1), (3, 8, 8, 2560, 0, 1280, 1),
          (1, 8, 8, 1280, 0, 1280, 1), (1, 12, 12, 1280, 0, 1280, 3)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B, h, w, c0, c1, n, ks in shapes:
    cin = c0 + c1
    pw = ops.to_device_pack(pack_conv(rnd(n, cin, ks, ks), rnd(n)))
    g = Geom.conv(h, w, ksize=ks, batch=B)
    x0 = rnd(g.m, c0).cuda(); x1 = rnd(g.m, c1).cuda() if c1 else None
    res = rnd(g.m, n).cuda()
    outs = {}
    row = []
    for name, kw in [("table", {}), ("stream", {"pipeline": 9, "tile": 2})]:
        out = torch.zeros(g.m, n, dtype=torch.float16, device="cuda")
        f = lambda: ops.conv(x0, x1, g, pw, out, c0=c0, c1=c1, act=2, residual=res, **kw)
        for _ in range(3): f()
        ops.synchronize()
        best = 1e9
        for _ in range(3):
            e0.record(ops.stream)
            for _ in range(30): f()
            e1.record(ops.stream); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
        outs[name] = out
        key = ops.conv_key(g, pw, 0, False)
        row.append(f"{name}: {best:6.1f} us ({n * cin * ks * ks * 2 / best / 1e6:5.2f} TB/s)" + (f" cfg={ops.tile_override.get(key)}" if name == "table" else ""))
    err = float((outs["table"].float() - outs["stream"].float()).abs().max())
    print(f"B={B} {h}x{w} cin={c0}+{c1} n={n} k={ks} M={g.m} W={n*cin*ks*ks*2/1e6:5.1f}MB | " + " | ".join(row) + f" | maxdiff {err:.2e}", flush=True)
