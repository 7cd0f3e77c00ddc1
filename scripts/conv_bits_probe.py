"""Do a conv's output BITS depend on the tile / pipeline at a fixed split-K?  (round 5: a twin pair of the UNet / ControlNet encoders
runs in a group form; if only split_k and the halo form move the bits, a pair restricted to its members' split gives the frame the
two-stream form gives.)   python scripts/conv_bits_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import lib as L  # noqa: E402
from videosd_amd.ops import Geom, HipOps  # noqa: E402
from videosd_amd.packing import pack_conv  # noqa: E402

ops = HipOps(0)
g = torch.Generator().manual_seed(0)
for (h, c, n, ks) in [(16, 1280, 1280, 3), (32, 640, 640, 1), (64, 320, 320, 3), (8, 2560, 1280, 1)]:
    x = (torch.randn(h * h, c, generator=g)).half().cuda()
    wt = (torch.randn(n, c, ks, ks, generator=g) * (ks * ks * c) ** -0.5).half()
    b = (torch.randn(n, generator=g) * 0.1).half()
    res = torch.randn(h * h, n, generator=g).half().cuda()
    pw = ops.to_device_pack(pack_conv(wt, b))
    geom = Geom.conv(h, h) if ks == 3 else Geom.linear(h * h)
    for sp in (1, 4):
        outs = {}
        for ink in (True, False):
            ops.inkernel_splitk = ink
            for t in (L.TILE_64x64, L.TILE_64x128, L.TILE_128x64, L.TILE_128x128):
                for pl in (0, 3, 4, 5, 6):
                    out = torch.zeros(h * h, n, dtype=torch.float16, device="cuda")
                    try:
                        ops.conv(x, None, geom, pw, out, residual=res, act=2, tile=t, split_k=sp, pipeline=pl)
                        ops.synchronize()
                    except RuntimeError:
                        continue
                    outs[(t, pl, ink)] = out.clone()
                if sp == 1:
                    pass
            if sp == 1:
                break
        keys = list(outs)
        ref = outs[keys[0]]
        diff = [k for k in keys if not torch.equal(outs[k], ref)]
        print(f"{h}x{h} {c}->{n} k{ks} split {sp}: {len(keys)} forms, {len(diff)} differ from {keys[0]}: {diff[:6]}", flush=True)
ops.inkernel_splitk = True
