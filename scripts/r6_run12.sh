#!/bin/bash
mkdir -p gpurun_out/r6
echo "== parity of the new forms"; timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -5
import torch, sys
sys.path.insert(0, ".")
from videosd_amd.ops import Geom, HipOps
from videosd_amd.packing import pack_conv
ops = HipOps(0)
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()
for (B, H, W, cin, cout, ks) in [(2, 32, 32, 128, 320, 3), (1, 40, 24, 640, 200, 1), (3, 16, 16, 320, 640, 3)]:
    pw = ops.to_device_pack(pack_conv(r(cout, cin, ks, ks), r(cout)))
    g = Geom.conv(H, W, ksize=ks, batch=B)
    x = r(g.m, cin).cuda(); res = r(g.m, cout).cuda()
    outs = []
    for pl in (3, 9, 11, 12):
        o = torch.zeros(g.m, cout, dtype=torch.float16, device="cuda")
        ops.conv(x, None, g, pw, o, residual=res, act=2, tile=0, split_k=1, pipeline=pl)
        ops.synchronize(); outs.append(o.cpu())
    print((B, H, W, cin, cout, ks), "bit-identical:", all(torch.equal(outs[0], o) for o in outs[1:]))
PY
echo "== probe lanes"; timeout 1500 python scripts/w8_probe.py --mode1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/w8x2_probe_mode1.txt
