"""Fused transformer tail (csrc/fused_tail.hip) against the launches it replaces, same weights: python scripts/tail_bench.py [M ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from videosd_amd.ops import HipOps, Geom
from test_ops_gpu import _tail_weights
ops = HipOps(0)
ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
c = 320
w, packs = _tail_weights(c)
pk = {k: ops.to_device_pack(v) for k, v in packs.items()}
def timeit(fn, n=50):
    for _ in range(5): fn()
    ops.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ops.stream)
    for _ in range(n): fn()
    e1.record(ops.stream); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for m in [int(a) for a in sys.argv[1:]] or [12288, 4096]:
    r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).half()
    att, h, x = r(m, c), r(m, c), r(m, c)
    h1, q, out, h2, f = (torch.zeros(m, c, dtype=torch.float16, device="cuda") for _ in range(5))
    ff = torch.zeros(m, 4 * c, dtype=torch.float16, device="cuda")
    rs1 = torch.zeros(m, c // 64, 2, dtype=torch.float32, device="cuda"); rs2 = torch.zeros_like(rs1)
    lin = Geom.linear(m)
    ta = timeit(lambda: ops.tail_a(att, h, m, pk["out1"], pk["q2"], h1, q))
    tb = timeit(lambda: ops.tail_b(att, h1, x, m, pk["out2"], pk["ff1"], pk["ff2"], pk["proj"], out))
    def unfused_a():
        ops.conv(att, None, lin, pk["out1"], h1, residual=h, rowstat_out=rs1)
        ops.conv(h1, None, lin, pk["q2"], q, ln_part=rs1)
    def unfused_b():
        ops.conv(att, None, lin, pk["out2"], h2, residual=h1, rowstat_out=rs2)
        ops.conv(h2, None, lin, pk["ff1"], ff, ln_part=rs2)
        ops.conv(ff, None, lin, pk["ff2"], f, residual=h2)
        ops.conv(f, None, lin, pk["proj"], out, residual=x)
    ua, ub = timeit(unfused_a), timeit(unfused_b)
    print(f"M={m}: tail_a {ta:.1f} us (2 launches: {ua:.1f})   tail_b {tb:.1f} us (4 launches: {ub:.1f})", flush=True)
