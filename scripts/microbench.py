import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.ops import HipOps, Geom
from videosd_amd.packing import pack_linear, pack_conv
from videosd_amd import lib as L
ops = HipOps(0)
def timeit(fn, n=200):
    for _ in range(5): fn()
    ops.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ops.stream)
    for _ in range(n): fn()
    e1.record(ops.stream); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = torch.zeros(8, dtype=torch.float16, device="cuda"); b = torch.zeros_like(a); o = torch.zeros_like(a)
print("trivial axpy (1 block): %.2f us" % timeit(lambda: ops.axpy(a, b, 1.0, 8, o)))
def rnd(*s): return (torch.randn(*s, device="cuda") * 0.1).half()
for (m, n, k, ks) in [(4096, 320, 320, 1), (4096, 320, 2880, 3), (1024, 640, 5760, 3), (256, 1280, 11520, 3)]:
    if ks == 1:
        x = rnd(m, k); pw = ops.to_device_pack(pack_linear(rnd(n, k).cpu(), rnd(n).cpu())); g = Geom.linear(m)
    else:
        hh = int(m ** 0.5); cin = k // 9
        x = rnd(m, cin); pw = ops.to_device_pack(pack_conv(rnd(n, cin, 3, 3).cpu(), rnd(n).cpu())); g = Geom.conv(hh, hh)
    out = torch.zeros(m, n, dtype=torch.float16, device="cuda"); res = rnd(m, n)
    print(f"\n({m},{n},{k}) ks={ks}  {2*m*n*k/1e9:.2f} GFLOP")
    for tile in (0, 1, 2, 3):
        for pl in (3, 5, 6):
            for sp in (1, 2):
                t = timeit(lambda: ops.conv(x, None, g, pw, out, tile=tile, split_k=sp, pipeline=pl, residual=res), 100)
                t2 = timeit(lambda: ops.conv(x, None, g, pw, out, tile=tile, split_k=sp, pipeline=pl), 100)
                print(f"   tile={tile} pipe={pl} split={sp}: {t:6.1f} us (no residual {t2:6.1f})  {2*m*n*k/t/1e6:6.1f} TF/s")
