"""The autotuner's candidate table for one conv shape: python scripts/tune_conv_shape.py B H W CIN COUT [KSIZE=3] [--mode1]
(--mode1: the throughput-mode choice -- the shortlist timed with four copies in flight on the four launch lanes)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.ops import HipOps, Geom
from videosd_amd.packing import pack_conv
argv = [x for x in sys.argv[1:] if not x.startswith('--')]
a = [int(x) for x in argv[:6]] + [3] * (6 - len(argv[:6]))
B, H, W, cin, cout, ks = a[:6]
ops = HipOps(0)
ops.tune_mode = 1 if '--mode1' in sys.argv else 0
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()
pw = ops.to_device_pack(pack_conv(r(cout, cin, ks, ks), r(cout)))
g = Geom.conv(H, W, ksize=ks, batch=B)
x = r(g.m, cin).cuda(); out = torch.zeros(g.m, cout, dtype=torch.float16, device="cuda"); rv = r(cout).cuda()
best, table = ops.tune_conv((x, None, g, pw, out), dict(rowvec=rv, act=2))
print(f"B={B} {H}x{W} {cin}->{cout} k={ks}: M={g.m}  flops={2.0*g.m*cout*cin*ks*ks/1e9:.1f} G")
for us, t, sp, ink, pl in table[:14]:
    print(f"   {us:7.1f} us  {2.0*g.m*cout*cin*ks*ks/us/1e6:6.0f} TF/s  tile={t} split={sp} ink={ink} pipe={pl}")
print(f"   ... {len(table)} candidates, worst {table[-1][0]:.1f} us")
