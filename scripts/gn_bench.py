"""GroupNorm(+SiLU) launch pair timings at the UNet's shapes (both kernels, back to back on one stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.ops import HipOps
ops = HipOps(0)
def timeit(fn, n=200):
    for _ in range(5): fn()
    ops.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ops.stream)
    for _ in range(n): fn()
    e1.record(ops.stream); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # images per launch
for hw, c in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1920), (256, 1280), (256, 2560), (64, 1280), (64, 2560), (16384, 320)]:
    x = (torch.randn(B * hw, c, device="cuda")).half(); g = torch.ones(c, device="cuda").half(); b = torch.zeros(c, device="cuda").half()
    o = torch.empty_like(x)
    t = timeit(lambda: ops.groupnorm(x, None, c, 0, hw, 32, 1e-5, g, b, True, o, batch=B))
    print(f"gn {B} x ({hw},{c}): {t:6.1f} us  ({B*hw*c*2*3/t/1e3:7.1f} GB/s algorithmic)")
