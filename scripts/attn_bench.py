"""Attention launches at the UNet's shapes, every launch shape (waves, query blocks per wave, key split) of
csrc/attention.hip: back-to-back launches on one stream, HIP events around 20 of them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videosd_amd.lib as _lib
if os.environ.get("ATTN_BENCH_LIB"):  # an experimental build of the library (development only)
    _lib.LIB_PATH = os.path.abspath(os.environ["ATTN_BENCH_LIB"])
from videosd_amd.ops import HipOps
ops = HipOps(0)
shapes = [(4096, 4096, 8, 40, 5), (4096, 4096, 8, 40, 3), (4096, 4096, 8, 40, 2), (4096, 4096, 8, 40, 1), (1024, 1024, 8, 80, 5), (1000, 1000, 8, 80, 3), (3000, 2999, 8, 40, 2), (256, 256, 8, 160, 5), (20480, 77, 8, 40, 1)] if os.environ.get("ATTN_BENCH_SHORT") else [(4096, 4096, 8, 40, 1), (1024, 1024, 8, 80, 1), (256, 256, 8, 160, 1), (4096, 77, 8, 40, 1), (1024, 77, 8, 80, 1),
          (4096, 4096, 10, 64, 1), (1024, 1024, 20, 64, 1), (4096, 4096, 8, 40, 2), (4096, 4096, 8, 40, 3), (1024, 1024, 8, 80, 3),
          (256, 256, 8, 160, 3), (9216, 9216, 8, 40, 1)]
variants = ["auto", "4,1,1", "4,1,2", "2,1,1", "2,1,2"] if os.environ.get("ATTN_BENCH_SHORT") else ["auto", "4,1,1", "4,1,2", "2,1,1"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for sq, sk, heads, d, B in shapes:
    c = heads * d
    t_img = (sk + 63) // 64 * 64
    torch.manual_seed(sq + d)
    q = torch.randn(B * sq, c, device="cuda").half(); k = torch.randn(B * sk, c, device="cuda").half()
    vt = torch.zeros(c, B * t_img, device="cuda").half()
    for b in range(B):
        vt[:, b * t_img:b * t_img + sk] = torch.randn(c, sk, device="cuda").half()
    o = torch.empty(B * sq, c, device="cuda").half()
    f = lambda: ops.attention(q, c, k, c, vt, B * t_img, o, c, sq, sk, heads, d, d ** -0.5, batch=B, k_brows=sk, vt_bcols=t_img)
    row = []
    ref = None
    for v in variants:
        if v == "auto":
            os.environ.pop("VSD_ATTN_SHAPE", None)
        else:
            os.environ["VSD_ATTN_SHAPE"] = v
        for _ in range(3): f()
        ops.synchronize()
        e0.record(ops.stream)
        for _ in range(20): f()
        e1.record(ops.stream); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        if ref is None: ref = o.clone()
        err = float((o.float() - ref.float()).abs().max())
        row.append(f"{v}: {us:6.1f}us" + ("" if err < 2e-3 else f" (maxdiff {err:.3g}!)"))
    os.environ.pop("VSD_ATTN_SHAPE", None)
    print(f"attn sq={sq} sk={sk} h={heads} d={d} B={B} [{4.0*sq*sk*heads*d*B/1e9:6.1f} GF] " + " | ".join(row) + f"  sum16={int(ref.view(torch.int16).long().sum())}", flush=True)
