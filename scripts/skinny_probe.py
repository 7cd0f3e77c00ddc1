"""Development: phase stamps (shader clock) of the weight-streaming kernel (csrc/conv_skinny.hip) built with -DSK_PROBE.
Build the probe library first (not part of libvsd.so):
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DSK_PROBE -c videosd_amd/csrc/conv_skinny.hip -o /tmp/skinny_probe.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build_exp/lib_skprobe.so $(ls videosd_amd/build/*.o | grep -v "conv_skinny.o\|probe") /tmp/skinny_probe.o
"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import videosd_amd.lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_exp", "lib_skprobe.so")
from videosd_amd.ops import HipOps, Geom
from videosd_amd.packing import pack_conv
ops = HipOps(0)
g_ = torch.Generator().manual_seed(1)
rnd = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()
for B, h, w, cin, n, ks in [(3, 8, 8, 1280, 1280, 3), (1, 8, 8, 1280, 1280, 3), (3, 8, 8, 1280, 1280, 1), (3, 8, 8, 5120, 1280, 1)]:
    pw = ops.to_device_pack(pack_conv(rnd(n, cin, ks, ks), rnd(n)))
    g = Geom.conv(h, w, ksize=ks, batch=B)
    x0 = rnd(g.m, cin).cuda()
    out = torch.zeros(g.m, n, dtype=torch.float16, device="cuda")
    S = cin // 128
    ws = torch.zeros(S * g.m * n + 64, dtype=torch.float32, device="cuda")
    for _ in range(5):
        ops.conv(x0, None, g, pw, out, act=2, pipeline=9, tile=2, workspace=ws)
    ops.synchronize()
    st = ws[S * g.m * n:].view(torch.int64).cpu().tolist()
    for name, a in (("first WG", st[0:7]), ("last WG", st[8:15])):
        d = [a[i + 1] - a[i] for i in range(6)]
        print(f"B={B} cin={cin} k={ks} {name}: issue loads {d[0]}, park panel (wait fill) {d[1]}, pb+zero {d[2]}, barrier {d[3]}, main loop {d[4]}, "
              f"reduce+store {d[5] if len(d) > 5 else 0}; total {a[6] - a[0]}", flush=True)
