"""Run under rocprofv3 --kernel-trace: executes eager frames of the 512x512 4-step program and writes the op
list (shapes, tile, split) to gpurun_out/ops_<tag>.json so that scripts/analyze_trace.py can attribute
kernel durations to layers.  --lanes: the THROUGHPUT-mode forms of every layer (what a coalesced launch on four busy lanes runs:
the table's mode-1 entries -- larger tiles, the halo form, the eight-wave forms), here one launch at a time."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W, lib as L
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps, choose_tile

cn = "--no-cn" not in sys.argv
size = 512
batch = 1
for a in sys.argv[1:]:
    if a.startswith("--size="): size = int(a.split("=")[1])
    if a.startswith("--batch="): batch = int(a.split("=")[1])
tag = sys.argv[-1] if not sys.argv[-1].startswith("--") and len(sys.argv) > 1 else "r1"
ops = HipOps(0)
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
eng.overlap_controlnet = False
eng.twin_encoders = False  # (a table of LAYERS: the two encoders' twin layers as launches of their own)
eng.group_shortcuts = False  # (... and a ResnetBlock's shortcut conv beside its conv1, not in its grid)
eng.tune_for_lanes = "--lanes" in sys.argv
eng.prepare(size, size, 4, 0.6, use_controlnet=cn, use_graph=False, batch=batch)
ops.tune_mode = 1 if eng.tune_for_lanes else 0
meta = []


def conv_meta(a, k):
    g, w = a[2], a[3]
    tile, split = k.get("tile"), k.get("split_k")
    if tile is None:
        tile, sk = choose_tile(g.m, w.n, w.kp, w.geglu, k.get("t_col0", 0) if k.get("out_t") is not None else 0)
        split = sk if split is None else split
    key = ops.conv_key_of(g, w, k)
    ink = True
    if k.get("tile") is None and key in ops.tile_override:
        tile, split, ink, _pl = ops.tile_override[key]
    split = split or 1
    kt = w.kp // 64
    split = min(split, kt); per = -(-kt // split); split = -(-kt // per)
    return dict(op="conv", M=g.m, N=w.n, K=w.k, ks=g.ksize, stride=g.stride, resize=(g.hi != g.hs), tile=tile, split=split if not ink else -split,
                flops=2.0 * g.m * w.n * w.k, wbytes=2 * w.n * w.kp, geglu=w.geglu)


for fn, a, k in eng.program.calls:
    name = fn.__name__
    if name in ("fork", "join", "use_stream", "signal", "wait"):
        continue
    m = {"op": name}
    if name == "conv":
        m = conv_meta(a, k)
    elif name == "conv_group":  # several independent convs in one grid (the ControlNet merges): one kernel, the members' sums
        ent = ops.tile_override.get(ops.group_key(a[0], k.get("split")))
        if ent is not None and ent[0] == ops.GROUP_ALONE:  # (the table sends this group's members out as launches of their own)
            meta.extend(conv_meta(aa, kk) for aa, kk in a[0])
            continue
        mem = [aa for aa, _kk in a[0]]
        m.update(members=len(mem), M=sum(aa[2].m for aa in mem), flops=sum(2.0 * aa[2].m * aa[3].n * aa[3].k for aa in mem),
                 wbytes=sum(2 * aa[3].n * aa[3].kp for aa in mem))
    elif name == "tail_a":
        m.update(M=a[2], flops=2.0 * a[2] * 2 * 320 * 320, wbytes=2 * 2 * 320 * 320)
    elif name == "tail_b":
        m.update(M=a[3], flops=2.0 * a[3] * (2 * 320 * 320 + 3 * 320 * 1280), wbytes=2 * (2 * 320 * 320 + 3 * 320 * 1280))
    elif name == "groupnorm":
        m.update(C=a[2] + a[3], hw=a[4])
    elif name == "layernorm":
        m.update(rows=a[1], C=a[2])
    elif name == "attention":
        m.update(sq=a[8], sk=a[9], heads=a[10], d=a[11], flops=4.0 * a[8] * a[9] * a[10] * a[11] * k.get("batch", 1))
    meta.append(m)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(meta, open(f"gpurun_out/ops_{tag}.json", "w"))
f = np.random.default_rng(0).integers(0, 256, (size, size, 3) if batch == 1 else (batch, size, size, 3), dtype=np.uint8)
for _ in range(4):
    eng.infer_u8(f)
ops.synchronize()
print("ops", len(meta))
