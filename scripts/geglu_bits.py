"""where do the GEGLU outputs of two kernel forms differ? (development probe)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.ops import Geom, HipOps
from videosd_amd.packing import pack_geglu
ops = HipOps(0)
g_ = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g_) * sc).half()
m, c, hid = 1000, 384, 256
x = r(m, c); wg, bg = r(2 * hid, c, sc=c ** -0.5), r(2 * hid, sc=0.1)
pg = ops.to_device_pack(pack_geglu(wg, bg))
outs = {}
for tl, pl in ((0, 3), (0, 5), (0, 8), (0, 9), (4, 3), (4, 5), (4, 8), (4, 9), (6, 8), (6, 9), (3, 3)):
    o = torch.zeros(m, hid, dtype=torch.float16, device="cuda")
    ops.conv(x.cuda(), None, Geom.linear(m), pg, o, tile=tl, split_k=1, pipeline=pl)
    ops.synchronize()
    outs[(tl, pl)] = o.cpu()
base = outs[(0, 3)]
for k, v in outs.items():
    d = (v.float() - base.float()).abs()
    nz = (d > 0).sum().item()
    print(k, "differing", nz, "max", d.max().item(), "rows", sorted(set((d > 0).nonzero()[:, 0].tolist()))[:8] if nz else "")
