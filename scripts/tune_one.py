"""Candidate table of the autotuner for the absorbed cross-attention GEMMs (or any linear shape): python scripts/tune_one.py M C"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import lib as L
from videosd_amd.ops import HipOps, Geom
from videosd_amd.packing import pack_cross_attention, pack_linear
m, c = int(sys.argv[1]), int(sys.argv[2])
ops = HipOps(0)
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.1).half()
xa1, xa2 = pack_cross_attention(r(77, c).float(), r(77, c).float(), r(c, c), r(c, c), r(c), torch.ones(c).half(), torch.zeros(c).half(), 8)
xa1, xa2 = ops.to_device_pack(xa1), ops.to_device_pack(xa2)
h = r(m, c).cuda(); rs = torch.zeros(m, c // 64, 2, dtype=torch.float32, device="cuda")
pr = torch.zeros(m, 1024, dtype=torch.float16, device="cuda"); out = torch.zeros(m, c, dtype=torch.float16, device="cuda")
rs2 = torch.zeros(m, c // 64, 2, dtype=torch.float32, device="cuda")
for name, args, kw in (("softmax GEMM", (h, None, Geom.linear(m), xa1, pr), dict(ln_part=rs, act=L.ACT_SOFTMAX, softmax_cols=77)),
                       ("output GEMM", (pr, None, Geom.linear(m), xa2, out), dict(residual=h, rowstat_out=rs2))):
    best, table = ops.tune_conv(args, kw)
    print(name, "M", m, "C", c)
    for us, t, sp, ink, pl in table[:8]:
        print(f"   {us:7.1f} us tile={t} split={sp} ink={ink} pipe={pl}")
    print(f"   ... {len(table)} candidates, worst {table[-1][0]:.1f}")
