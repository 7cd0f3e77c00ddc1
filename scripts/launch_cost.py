import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps
ops = HipOps(0)
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.overlap_controlnet = False
eng.prepare(512, 512, 4, 0.6, use_controlnet=True)
for _ in range(3): eng.launch()
ops.synchronize()
ts = []
for _ in range(10):
    t = time.perf_counter(); eng.launch(); ts.append(time.perf_counter() - t); ops.synchronize()
print("host time of one graph launch (GPU idle): %.3f ms" % (np.median(ts) * 1e3))
t = time.perf_counter()
for _ in range(10): eng.launch()
t1 = time.perf_counter() - t
ops.synchronize()
t2 = time.perf_counter() - t
print("10 back-to-back launches: host enqueue %.2f ms, total %.2f ms" % (t1 * 1e3, t2 * 1e3))
