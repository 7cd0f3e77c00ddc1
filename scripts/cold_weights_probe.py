"""Does a small-image layer run slower when its weights come from HBM than when they sit in the memory-side cache?

python scripts/cold_weights_probe.py [B=1]

For each shape: the tuned kernel form is timed (a) back to back on ONE weight tensor (what the tuner sees: the weights
stay in the 256 MB memory-side cache) and (b) cycling over enough copies of the weights to exceed it (what a frame
sees: 2.4 GB of weights per denoising step).  (c) = (b) with a touch kernel reading the NEXT copy on a second stream."""
import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.ops import HipOps, Geom
from videosd_amd.packing import pack_conv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ops = HipOps(0)
g_ = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g_) * 0.05).half()
SHAPES = [(8, 8, 1280, 1280, 3), (8, 8, 2560, 1280, 3), (16, 16, 1280, 1280, 3), (16, 16, 2560, 1280, 3), (16, 16, 1280, 1280, 1),
          (16, 16, 1280, 10240, 1), (16, 16, 5120, 1280, 1), (32, 32, 640, 640, 3), (32, 32, 1280, 640, 3), (32, 32, 640, 640, 1)]
for (H, W, cin, cout, ks) in SHAPES:
    pw = ops.to_device_pack(pack_conv(r(cout, cin, ks, ks), r(cout)))
    wbytes = pw.weight.numel() * 2
    ncopy = max(2, min(300, (600 << 20) // wbytes))
    copies = []
    for _ in range(ncopy):
        q = copy.copy(pw)
        q.weight = pw.weight.clone()  # its own copy of the big tensor
        copies.append(q)
    g = Geom.conv(H, W, ksize=ks, batch=B)
    x = r(g.m, cin).cuda(); out = torch.zeros(g.m, cout, dtype=torch.float16, device="cuda"); rv = r(cout).cuda()
    best, table = ops.tune_conv((x, None, g, pw, out), dict(rowvec=rv, act=2))

    def run(wt):
        ops.conv(x, None, g, wt, out, rowvec=rv, act=2)

    def timeit(fn, n):
        for i in range(4): fn(i)
        ops.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ops.stream)
        for i in range(n): fn(i)
        e1.record(ops.stream); e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    warm = timeit(lambda i: run(pw), max(60, 2 * ncopy))
    cold = timeit(lambda i: run(copies[i % ncopy]), max(60, 2 * ncopy))

    # (c) does touching one dword of every 128-byte line of the NEXT copy (on the same stream, before this layer) make
    # the next layer run as on warm weights?  (d) = the same touch kernel on an unrelated buffer: what the touch itself costs.
    other = [c.weight.clone() for c in copies[:max(2, ncopy // 2)]]

    def touch_next(i):
        copies[(i + 1) % ncopy].weight.view(torch.int32).view(-1)[::32].sum()
        run(copies[i % ncopy])

    def touch_other(i):
        other[i % len(other)].view(torch.int32).view(-1)[::32].sum()
        run(copies[i % ncopy])
    with torch.cuda.stream(ops.stream):
        pre = timeit(touch_next, max(60, 2 * ncopy))
        base = timeit(touch_other, max(60, 2 * ncopy))
    print(f"B={B} {H}x{W} {cin}->{cout} k={ks}: weights {wbytes/1e6:5.1f} MB x {ncopy}; form {best[1:]}; "
          f"warm {warm:6.1f} us  cold {cold:6.1f} us | touch next + layer {pre:6.1f} us, touch other + layer {base:6.1f} us "
          f"-> prefetched layer ~ {cold - (base - pre):6.1f} us", flush=True)
    del other
    del copies
    torch.cuda.empty_cache()
