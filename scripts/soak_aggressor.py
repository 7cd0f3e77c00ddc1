"""Which concurrent work disturbs the Sobel control image? One engine replays its graph (one launch in flight) while a
second stream runs an 'aggressor': torch matmuls, torch elementwise, or a second engine without ControlNet."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps
kind = sys.argv[1] if len(sys.argv) > 1 else "mm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ops = HipOps(0)
ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.overlap_controlnet = False
eng.prepare(512, 512, 4, 0.6, use_controlnet=True, batch=1)
other = None
if kind in ("nocn", "cn"):
    other = eng.make_slot(); other.prepare(512, 512, 4, 0.6, use_controlnet=(kind == "cn"), batch=1)
conv_job = None
if kind.startswith("conv"):   # conv:<pipeline>:<h>:<cin>:<cout>:<ksize>
    from videosd_amd.ops import Geom
    from videosd_amd.packing import pack_conv
    f = [int(v) for v in kind.split(":")[1:]]
    pl, ch, cin, cout, ks = f[:5]; tl = f[5] if len(f) > 5 else None
    other = eng.make_slot()
    pw = pack_conv(torch.randn(cout, cin, ks, ks) * 0.05, torch.zeros(cout)); pw.weight = pw.weight.cuda(); pw.bias = pw.bias.cuda()
    g = Geom.conv(ch, ch, ksize=ks)
    xs = torch.randn(g.m, cin, device="cuda").half(); out = torch.zeros(g.m, cout, dtype=torch.float16, device="cuda")
    def conv_job():
        for _ in range(30):
            other.ops.conv(xs, None, g, pw, out, ldo=cout, c0=cin, c1=0, pipeline=(None if pl < 0 else pl), tile=tl,
                           split_k=(1 if tl is not None else None))
rng = np.random.default_rng(0)
inputs = [rng.integers(0, 256, (512, 512, 3), dtype=np.uint8) for _ in range(4)]
ref_ctrl, ref_x0, ref_out = [], [], []
for x in inputs:
    ref_out.append(eng.infer_u8(x).copy()); ref_ctrl.append(eng.buffers["control"].clone()); ref_x0.append(eng.buffers["x0"].clone())
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.half); b = torch.randn(4096, 4096, device="cuda", dtype=torch.half)
v = torch.randn(64 << 20, device="cuda")
bad = bad_x0 = bad_out = 0
for i in range(n):
    k = i % 4
    if kind == "mm":
        with torch.cuda.stream(side):
            for _ in range(40): torch.mm(a, b)
    elif kind == "ew":
        with torch.cuda.stream(side):
            for _ in range(60): v.mul_(1.0001)
    elif conv_job is not None:
        conv_job()
    elif other is not None:
        other.submit_u8(inputs[(k + 1) % 4])
    eng.submit_u8(inputs[k]); out_k = eng.collect_u8()
    if other is not None and conv_job is None: other.collect_u8()
    if conv_job is not None: other.ops.synchronize() if hasattr(other.ops, "synchronize") else torch.cuda.synchronize()
    if not torch.equal(eng.buffers["control"], ref_ctrl[k]): bad += 1
    if not torch.equal(eng.buffers["x0"], ref_x0[k]): bad_x0 += 1
    if not np.array_equal(out_k, ref_out[k]): bad_out += 1
torch.cuda.synchronize()
print(f"aggressor={kind}: of {n} launches, wrong control image {bad}, wrong encoded latent {bad_x0}, wrong output frame {bad_out}", flush=True)
