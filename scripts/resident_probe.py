"""Development: per-phase shader-clock totals of conv_resident_kernel built with -DRS_PROBE (build_exp/lib_rsprobe.so:
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DRS_PROBE -c videosd_amd/csrc/conv_resident.hip -o /tmp/res_probe.o;
hipcc --offload-arch=gfx950 -shared -fPIC -o build_exp/lib_rsprobe.so $(ls videosd_amd/build/*.o | grep -v "conv_resident.o\\|probe") /tmp/res_probe.o)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import videosd_amd.lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_exp", "lib_rsprobe.so")
from videosd_amd.ops import HipOps, Geom
from videosd_amd.packing import pack_conv
ops = HipOps(0)
g_ = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()
pw = ops.to_device_pack(pack_conv(rnd(64, 64, 3, 3), rnd(64)))
for B, h, w in [(5, 512, 512), (1, 512, 512)]:
    g = Geom.conv(h, w, batch=B)
    x = rnd(B * h * w, 64).cuda(); res = rnd(g.m, 64).cuda()
    out = torch.zeros(g.m, 64, dtype=torch.float16, device="cuda")
    ws = torch.zeros(64, dtype=torch.float32, device="cuda")
    for _ in range(3):
        ops.conv(x, None, g, pw, out, act=1 | 256, residual=res, pipeline=10, tile=1, workspace=ws)
    ops.synchronize()
    st = ws.view(torch.int64)[:8].cpu().tolist()
    n = -(-B * (h // 8) * (w // 16) // 512)
    print(f"B={B} {h}x{w}: {n} patches per workgroup; cycles per patch: barrier(top) {st[1]//n}, issue halo DMA + residual loads {st[2]//n}, "
          f"taps x MFMA {st[3]//n}, wait halo/residual {st[4]//n}, epilogue {st[5]//n}; prologue (weights) {st[0]}", flush=True)
