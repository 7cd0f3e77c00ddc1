"""The whole per-frame program with EVERY device buffer -- packed weights, every activation tensor (the arena allocates tensor by
tensor), workspaces, frame / output buffers -- ending at (or starting after) an unmapped page (scripts/guard_pages.cpp through the
`HipOps.allocator` hook): an out-of-bounds access anywhere in the recorded program takes a GPU memory fault.  Ragged and
degenerate frame sizes, 1-3 frames per launch, SD1.5 + ControlNet and the mini SDXL topology; every result must equal the
ordinary engine's bit for bit.
usage (GPU box): python scripts/guard_page_engine.py [table] [lanes] [sdxl] [b=1,3] [sizes like 8x8 24x40 ...]
(table: the shipped tuning table's kernel forms instead of the heuristic's; lanes: its throughput-mode entries)"""
import ctypes as C
import os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import config as Cfg, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps

so = os.path.join(ROOT, "scripts", "libguardpages.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "scripts", "guard_pages.cpp"), "-o", so])
plain = HipOps(0)
gp = C.CDLL(so)
gp.guard_alloc.restype = C.c_void_p
gp.guard_alloc.argtypes = [C.c_size_t, C.c_int]
rng = np.random.default_rng(0)
mapped = [0]


class _Iface:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def guard_allocator(nbytes):
    ptr = gp.guard_alloc(int(nbytes), int(rng.random() < 0.75))
    assert ptr, "guard_alloc failed"
    mapped[0] += nbytes
    return torch.as_tensor(_Iface(ptr, int(nbytes)), device="cuda")


def engines(kind):
    if kind == "sd15":
        ucfg, ccfg = Cfg.SD15_UNET, Cfg.SD15_CONTROLNET
    elif kind == "sdxl":
        ucfg, ccfg = Cfg.SDXL_UNET, None
    else:
        ucfg, ccfg = Cfg.MINI_SDXL_UNET, None
    wu = W.synthesize(W.unet_spec(ucfg), "unet.", device="cuda")
    wc = W.synthesize(W.controlnet_spec(ccfg), "cn.", device="cuda") if ccfg is not None else None
    wv = W.synthesize(W.taesd_spec(Cfg.TAESD), "vae.", device="cuda")
    out = []
    for hook in (None, guard_allocator):
        ops = plain if hook is None else HipOps(0, make_current=False)
        ops.allocator = hook
        if "table" in sys.argv:
            ops.load_tuning(os.path.join(ROOT, "profiles", "tuning_mi355x.json"))
        e = Engine(ops, ucfg, ccfg, Cfg.TAESD, wu, wc, wv)
        e.set_text_embeds((torch.randn(77, ucfg.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5).half())
        e.overlap_controlnet = False
        e.tune_for_lanes = "lanes" in sys.argv
        out.append(e)
    return out, ccfg is not None


BATCHES = [int(x) for a in sys.argv[1:] if a.startswith("b=") for x in a[2:].split(",")] or [1, 3]
sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if "x" in a and a[0].isdigit()] or [(8, 8), (16, 8), (24, 40), (72, 40), (104, 88), (200, 136), (360, 640)]
for kind in (("sdxl",) if "sdxl" in sys.argv else ("sd15", "mini")):
    (ref, grd), cn = engines(kind)
    for (h, w) in sizes:
        for b in BATCHES:
            if h * w * b > 512 * 512 * 5 and kind != "sdxl":
                continue
            f = rng.integers(0, 256, (h, w, 3) if b == 1 else (b, h, w, 3), dtype=np.uint8)
            print(f"{kind} {h}x{w} x{b}", flush=True)
            outs = []
            for e in (ref, grd):
                if kind in ("mini", "sdxl"):
                    e.set_added_cond(torch.full((e.ucfg.add_pooled_dim,), 0.25).half(), (h, w, 0, 0, h, w))
                e.prepare(h, w, 2, 0.6, use_controlnet=cn, use_graph=False, batch=b, autotune=False)
                outs.append(e.infer_u8(f).copy())
            assert np.array_equal(outs[0], outs[1]), f"{kind} {h}x{w} x{b}: guarded run differs from the ordinary one"
    if kind == "sd15":  # the reference-only program (banked self-attention keys / values + AdaIN): kernels nothing else runs
        for (h, w) in ((64, 64), (128, 192)):
            f, rf = (rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(2))
            print(f"{kind} reference-only {h}x{w}", flush=True)
            outs = []
            for e in (ref, grd):
                e.prepare(h, w, 2, 0.6, use_controlnet=False, use_graph=False, batch=1, ref_mode=True, autotune=False)
                e.ops.upload(e.ref_u8, torch.from_numpy(rf))
                outs.append(e.infer_u8(f).copy())
            assert np.array_equal(outs[0], outs[1]), f"reference-only {h}x{w}: guarded run differs from the ordinary one"
    print(f"{kind}: ok ({mapped[0] / 2**30:.1f} GB mapped so far)", flush=True)
# the text towers (once per prompt: CLIP-L, and the SDXL pair with the erf-GELU tower and its pooled row)
from videosd_amd import clip as K
for c1, c2 in ((Cfg.MINI_CLIP, K.MINI_CLIP_G), (Cfg.CLIP_L, None)) + (((K.SDXL_CLIP_L, K.SDXL_CLIP_G),) if "sdxl" in sys.argv else ()):
    got = []
    for hook in (None, guard_allocator):
        ops = plain if hook is None else HipOps(0, make_current=False)
        ops.allocator = hook
        t1 = K.ClipTextEncoder(ops, c1, W.synthesize(K.text_tower_spec(c1), "t1.", device="cuda"))
        ids = torch.randint(1, c1.vocab - 1, (77,), generator=torch.Generator().manual_seed(3))
        ids[9:] = c1.vocab - 1
        r = [t1.encode_ids(ids).float().cpu()]
        if c2 is not None:
            t2 = K.ClipTextEncoder(ops, c2, W.synthesize(K.text_tower_spec(c2), "t2.", device="cuda"))
            ids2 = ids.clone()
            ids2[10:] = 0
            e, pooled = K.SdxlTextEncoders(t1, t2).encode_ids(ids, ids2)
            r += [e.float().cpu(), pooled.float().cpu()]
        got.append(r)
    assert all(torch.equal(a, b) for a, b in zip(*got)), "text towers: guarded run differs"
    print(f"text towers width {c1.width}{' + ' + str(c2.width) if c2 else ''}: ok", flush=True)
print("guard page engine run passed")
