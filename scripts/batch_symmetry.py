"""Two IDENTICAL frames in one launch must come out identical: image b of a batch is computed by the same arithmetic as image 0
(every kernel treats rows independently or per image).  Runs the recorded program eagerly op by op and, after each op, compares
the two images' halves of every 2-D fp16 tensor the op touched; prints the first ops whose output halves differ.
usage: python scripts/batch_symmetry.py H W [steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps

H, Wd = int(sys.argv[1]), int(sys.argv[2])
ops = HipOps(0)
ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.overlap_controlnet = False
eng.prepare(H, Wd, int(sys.argv[3]) if len(sys.argv) > 3 else 2, 0.6, use_controlnet=True, use_graph=False, batch=2, autotune=False)
f = np.random.default_rng(0).integers(0, 256, (H, Wd, 3), dtype=np.uint8)
ops.upload(eng.frame_u8, torch.from_numpy(np.stack([f, f])))
bad = 0
known = set()
for i, (fn, a, k) in enumerate(eng.program.calls):
    fn(*a, **k)
    torch.cuda.synchronize()
    ts = [x for x in list(a) + list(k.values()) if isinstance(x, torch.Tensor) and x.dtype == torch.float16 and x.dim() == 2 and x.shape[0] % 2 == 0]
    for t in ts:
        if t.data_ptr() in known:
            continue
        h = t.shape[0] // 2
        if not torch.equal(t[:h], t[h:]):
            known.add(t.data_ptr())
            d = (t[:h].float() - t[h:].float()).abs()
            desc = fn.__name__
            if desc == "conv":
                g, w = a[2], a[3]
                desc += f" M={g.m} N={w.n} K={w.k} ks={g.ksize} stride={g.stride} kwargs={sorted(k)} cfg={ops.tile_override.get(ops.conv_key_of(g, w, k))}"
            print(f"op {i} {desc}: tensor {tuple(t.shape)} halves differ: max {float(d.max()):.4g}, {int((d > 0).sum())} elements", flush=True)
            bad += 1
    if bad >= 6:
        break
out = eng.out_u8.cpu().numpy()
print("frames equal:", np.array_equal(out[0], out[1]), "ops run:", i + 1, "of", len(eng.program.calls))
