"""Where does a conv launch's time go?  Workgroup timeline of one eager pass of the 512x512 4-step program.

Needs the instrumented build (python -m videosd_amd.build --timeline -> videosd_amd/libvsd_tl.so, -DVSD_WG_TIMELINE):
every workgroup of conv_gemm_kernel / conv_halo_kernel logs start, main-loop-done and end time (s_memrealtime, 10 ns
ticks) and its placement (HW_ID, XCC_ID).  Per distinct layer shape this prints: launches, grid, span of the launch
(first workgroup start -> last workgroup end), dispatch skew (first -> last START), workgroup life (median / p90), the
share of a life spent before the epilogue, CUs used and the most workgroups one CU ran, and the gap to the next
instrumented launch of the stream.

usage (GPU box): VSD_LIB=videosd_amd/libvsd_tl.so python scripts/wg_timeline.py [--batch=5] [--no-cn] [tag]
"""
import collections
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("VSD_LIB"), "set VSD_LIB=videosd_amd/libvsd_tl.so (python -m videosd_amd.build --timeline)"
from videosd_amd import config as Cfg, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402

batch, cn, tag = 5, True, "tl"
for a in sys.argv[1:]:
    if a.startswith("--batch="):
        batch = int(a.split("=")[1])
    elif a == "--no-cn":
        cn = False
    elif not a.startswith("--"):
        tag = a
ops = HipOps(0)
lib = ops.ctx.lib
lib.vsd_wgtl_set.argtypes = [C.c_void_p, C.c_int64]
lib.vsd_wgtl_set.restype = None
lib.vsd_wgtl_used.restype = C.c_int64
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ops.load_tuning(os.path.join(root, "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(Cfg.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(Cfg.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(Cfg.TAESD), "vae.", device="cuda")
eng = Engine(ops, Cfg.SD15_UNET, Cfg.SD15_CONTROLNET, Cfg.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.overlap_controlnet = False  # one stream: launch order = log order
eng.prepare(512, 512, 4, 0.6, use_controlnet=cn, use_graph=False, batch=batch)
f = np.random.default_rng(0).integers(0, 256, (512, 512, 3) if batch == 1 else (batch, 512, 512, 3), dtype=np.uint8)
for _ in range(3):
    eng.infer_u8(f)
ops.synchronize()
words = 64 << 20
log = torch.zeros(words, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
lib.vsd_wgtl_set(C.c_void_p(log.data_ptr()), words)
eng.infer_u8(f)
ops.synchronize()
torch.cuda.synchronize()
used = int(lib.vsd_wgtl_used())
lib.vsd_wgtl_set(None, 0)
host = log[:used].cpu().numpy().astype(np.uint64)

# the conv calls of the program, in launch order (pipelines 9 / 10 are not instrumented and not used by default)
convs = []
for fn, a, k in Engine.flat_calls(eng.program.calls):
    if fn.__name__ != "conv":
        continue
    g, w = a[2], a[3]
    key = ops.conv_key_of(g, w, k)
    cfg = ops.tile_override.get(key)
    kind = ("geglu" if w.geglu else "softmax" if w.tile128 else "qkv" if k.get("out_t") is not None else
            "ln" if k.get("ln_part") is not None else "plain")
    convs.append(dict(M=g.m, N=w.n, K=w.k, ks=g.ksize, cfg=cfg, epi=kind, flops=2.0 * g.m * w.n * w.k))

pos, regions = 0, []
while pos < used:
    grid = int(host[pos])
    if grid == 0:  # header not written: should not happen (every launch's workgroup 0 writes it)
        raise SystemExit(f"log broken at word {pos}")
    regions.append((grid, int(host[pos + 1]), host[pos + 2:pos + 2 + 8 * grid].reshape(grid, 8)))
    pos += 2 + 8 * grid
print(f"{len(regions)} instrumented launches logged, {len(convs)} conv calls in the program", flush=True)
assert len(regions) == len(convs), "log / program mismatch"

agg = collections.OrderedDict()
rows = []
for i, ((grid, kind, r), m) in enumerate(zip(regions, convs)):
    t0, t1, t2 = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 2].astype(np.int64)
    ok = t2 > 0
    hw = r[:, 3]
    hwid, xcc = (hw & np.uint64(0xffffffff)).astype(np.int64), ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    cu = ((hwid >> 8) & 0xf) | (((hwid >> 12) & 0x1) << 4) | (((hwid >> 13) & 0x7) << 5) | (xcc << 8)  # cu, sh, se, xcc
    base = t0[ok].min()
    span = (t2[ok].max() - base) / 100.0
    skew = (t0[ok].max() - base) / 100.0
    life = (t2[ok] - t0[ok]) / 100.0
    pre = np.where(t1[ok] > 0, (t1[ok] - t0[ok]) / np.maximum(t2[ok] - t0[ok], 1), 0.0)
    ta, tb, tc = (r[:, k].astype(np.int64)[ok] for k in (4, 5, 6))
    seg = lambda x, y: float(np.median(np.where((x > 0) & (y > 0), (y - x) / 100.0, 0.0)))  # noqa: E731
    # prologue (start -> first wait of the main loop), main loop, accumulators -> LDS + barrier, wait for residual / bias, walk + stores
    segs = dict(s_pro=seg(t0[ok], tc), s_loop=seg(np.where(tc > 0, tc, t0[ok]), t1[ok]), s_tr=seg(t1[ok], ta), s_res=seg(ta, tb), s_st=seg(np.where(tb > 0, tb, t1[ok]), t2[ok]))
    cnt = collections.Counter(cu[ok].tolist())
    gap = None
    if i + 1 < len(regions):
        nxt = regions[i + 1][2]
        gap = (int(nxt[:, 0].astype(np.int64)[nxt[:, 2] > 0].min()) - int(t2[ok].max())) / 100.0
    row = dict(i=i, grid=grid, kind=kind, span=span, skew=skew, life50=float(np.median(life)), life90=float(np.percentile(life, 90)),
               lifemin=float(life.min()), pre=float(np.median(pre)), cus=len(cnt), maxper=max(cnt.values()), gap=gap, **segs, **m)
    rows.append(row)
    key = (m["M"], m["N"], m["K"], m["ks"], m["epi"], "halo" if kind == 1 else "gemm", str(m["cfg"]), grid)
    agg.setdefault(key, []).append(row)

os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(f"gpurun_out/wgtl_{tag}_b{batch}.json", "w"))
tot = sum(r["span"] for r in rows)
print(f"sum of launch spans {tot/1e3:.2f} ms; sum of gaps to the next conv launch {sum(r['gap'] or 0 for r in rows)/1e3:.2f} ms (includes the other kernels between them)")
print(f"{'M':>7s} {'N':>6s} {'K':>6s} ks {'kind':8s} {'form':5s} {'cfg':22s} {'grid':>5s} {'cnt':>4s} {'span':>7s} {'skew':>6s} {'life50':>7s} {'life90':>7s} "
      f"{'min':>6s} {'pre%':>5s} {'CUs':>4s} {'max/CU':>6s} {'TF/s':>6s} {'tot ms':>7s} | {'prol':>5s} {'loop':>6s} {'a->lds':>6s} {'res':>5s} {'store':>5s}")
for key, rs in sorted(agg.items(), key=lambda kv: -sum(r["span"] for r in kv[1])):
    av = lambda f: sum(r[f] for r in rs) / len(rs)  # noqa: E731
    print(f"{key[0]:7d} {key[1]:6d} {key[2]:6d} {key[3]:2d} {key[4]:8s} {key[5]:5s} {key[6]:22s} {key[7]:5d} {len(rs):4d} {av('span'):7.1f} {av('skew'):6.1f} "
          f"{av('life50'):7.1f} {av('life90'):7.1f} {av('lifemin'):6.1f} {100*av('pre'):5.0f} {av('cus'):4.0f} {av('maxper'):6.1f} "
          f"{rs[0]['flops']/av('span')/1e6:6.0f} {sum(r['span'] for r in rs)/1e3:7.3f} | {av('s_pro'):5.1f} {av('s_loop'):6.1f} {av('s_tr'):6.1f} {av('s_res'):5.1f} {av('s_st'):5.1f}")
