"""What each kernel family OCCUPIES of the chip while the timed program runs.

`bench.py`'s `value` is 5 frames per launch x 4 launch lanes with the kernel forms that cost least on a shared chip (tuning mode 1);
its `roofline` block times each layer's fastest-ALONE form (mode 0) in a single-stream eager pass.  This script measures the timed
program itself: with the instrumented library (python -m videosd_amd.build --timeline -> libvsd_tl.so) every workgroup of every
kernel adds its life (first instruction -> exit, s_memrealtime) to a per-family counter, under captured-graph replay on all lanes.

    VSD_LIB=videosd_amd/libvsd_tl.so python scripts/wg_cu_time.py [--batch 5] [--seconds 2.0] [--out profiles/round5_wg_cu_time_5x4.txt]

Four runs of the same 512x512 4-step ControlNet program: kernel forms of mode 0 / mode 1, on one lane / on four lanes.  Per family
and frame: workgroup-milliseconds (the sum of workgroup lives), wave-milliseconds (lives x waves per workgroup), workgroups,
and the algorithmic FLOP per workgroup-second; per run: frames/s, the chip's mean resident waves per SIMD (wave-ms / (wall x 1024
SIMDs)).  Prints a table and one JSON line; --out writes both to a file."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("VSD_LIB"), "set VSD_LIB=videosd_amd/libvsd_tl.so (python -m videosd_amd.build --timeline)"
from videosd_amd import config as Cfg, weights as W  # noqa: E402
from videosd_amd import lib as L  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402

argv = sys.argv[1:]


def arg(name, default, cast=float):
    return cast(argv[argv.index(name) + 1]) if name in argv else default


B = arg("--batch", 5, int)
seconds = arg("--seconds", 2.0)
out_path = arg("--out", None, str)
FAMS = ["conv_gemm", "conv_halo", "splitk_reduce", "groupnorm", "attention", "fused_tail", "(unused)", "(unused2)"]
NF = len(FAMS)

ops = HipOps(0)
lib = ops.ctx.lib
lib.vsd_cut_set.argtypes = [C.c_void_p]
lib.vsd_cut_set.restype = None
cut_buf = torch.zeros(3 * 8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
lib.vsd_cut_set(C.c_void_p(cut_buf.data_ptr()))  # (read when a launch is ENQUEUED or captured: set before the plans are prepared)
ops.load_tuning(os.path.join(ROOT, "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(Cfg.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(Cfg.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(Cfg.TAESD), "vae.", device="cuda")
eng = Engine(ops, Cfg.SD15_UNET, Cfg.SD15_CONTROLNET, Cfg.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
engines = [eng] + [eng.make_slot() for _ in range(3)]
frames = np.random.default_rng(0).integers(0, 256, (B, 512, 512, 3) if B > 1 else (512, 512, 3), dtype=np.uint8)


def cut_read(reset):
    torch.cuda.synchronize()
    out = cut_buf.cpu().numpy().astype(np.float64).reshape(NF, 3)
    if reset:
        cut_buf.zero_()
        torch.cuda.synchronize()
    return out


def family_flops(e):
    """algorithmic FLOP of one launch of e's program, by family"""
    fl = dict.fromkeys(FAMS, 0.0)
    for fn, a, k in Engine.flat_calls(e.program.calls):
        name = fn.__name__
        if name == "conv":
            g, w = a[2], a[3]
            ent = e.ops.tile_override.get(e.ops.conv_key_of(g, w, k))
            halo = ent is not None and ent[3] in (7, 10)
            fl["conv_halo" if halo else "conv_gemm"] += 2.0 * g.m * w.n * w.k
        elif name == "attention":
            sq, sk, heads, d = a[8], a[9], a[10], a[11]
            fl["attention"] += 4.0 * sq * sk * heads * d * max(1, k.get("batch", 1))
        elif name == "tail_a":
            fl["fused_tail"] += 2.0 * a[2] * (2 * 320 * 320)
        elif name == "tail_b":
            fl["fused_tail"] += 2.0 * a[3] * (2 * 320 * 320 + 3 * 320 * 1280)
    return fl


def run(mode, lanes):
    pool = engines[:lanes]
    for e in pool:
        e.tune_for_lanes = bool(mode)
        e.overlap_launch = False
        e.prepare(512, 512, 4, 0.6, controlnet_scale=1.0, use_controlnet=True, batch=B, use_graph=True)
        e.infer_u8(frames)
    for e in pool:
        e.ops.synchronize()
    torch.cuda.synchronize()
    cut_read(True)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds or n % lanes:
        pool[n % lanes].launch(overlap=False)
        n += 1
        if n % (4 * lanes) == 0:  # (keep the host a few launches ahead, not thousands)
            pool[(n - 2 * lanes) % lanes].ops.synchronize()
    for e in pool:
        e.ops.synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    c = cut_read(True)
    nf = n * B
    fl = family_flops(pool[0])
    per = {}
    for i, f in enumerate(FAMS):
        wg_ms = c[i, 0] * 1e-5 / nf
        per[f] = {"wg_ms_per_frame": round(wg_ms, 4), "wave_ms_per_frame": round(c[i, 2] * 1e-5 / nf, 4), "workgroups_per_frame": round(c[i, 1] / nf, 1),
                  "tflop_per_wg_second": round(fl[f] / B / (wg_ms * 1e-3) / 1e12, 3) if wg_ms > 0 and fl[f] > 0 else None}
    wave_ms = sum(v["wave_ms_per_frame"] for v in per.values())
    return {"mode": mode, "lanes": lanes, "frames_per_launch": B, "launches": n, "fps": round(nf / wall, 2), "wall_ms_per_frame": round(wall / nf * 1e3, 4),
            "wg_ms_per_frame_total": round(sum(v["wg_ms_per_frame"] for v in per.values()), 3), "wave_ms_per_frame_total": round(wave_ms, 3),
            "mean_resident_waves_per_simd": round(wave_ms / (wall / nf * 1e3) / 1024.0, 3), "families": per}


runs = [run(0, 1), run(1, 1), run(0, 4), run(1, 4)]
lines = []
hdr = f"{'family':14s}" + "".join(f" | mode {r['mode']} x {r['lanes']} lane(s): wg-ms  wave-ms   WGs  TF/wg-s" for r in runs)
lines.append(f"512x512 4-step + ControlNet, {B} frame(s) per launch, captured graphs, {seconds:.1f} s per run; per FRAME")
lines.append(hdr)
for f in FAMS:
    row = f"{f:14s}"
    for r in runs:
        v = r["families"][f]
        tf = f"{v['tflop_per_wg_second']:8.2f}" if v["tflop_per_wg_second"] is not None else "       -"
        row += f" | {'':21s}{v['wg_ms_per_frame']:7.2f} {v['wave_ms_per_frame']:8.2f} {v['workgroups_per_frame']:6.0f} {tf}"
    lines.append(row)
row = f"{'TOTAL':14s}"
for r in runs:
    row += f" | {'':21s}{r['wg_ms_per_frame_total']:7.2f} {r['wave_ms_per_frame_total']:8.2f} {'':6s} {'':8s}"
lines.append(row)
for r in runs:
    lines.append(f"mode {r['mode']} x {r['lanes']}: {r['fps']:.1f} frames/s, wall {r['wall_ms_per_frame']:.3f} ms per frame, workgroup-ms per frame {r['wg_ms_per_frame_total']:.1f} "
                 f"(= {r['wg_ms_per_frame_total'] / r['wall_ms_per_frame']:.0f} workgroups resident on average; 256 CUs), mean resident waves per SIMD "
                 f"{r['mean_resident_waves_per_simd']:.2f}")
text = "\n".join(lines)
print(text)
js = json.dumps({"runs": runs})
print(js)
if out_path:
    with open(out_path, "w") as f:
        f.write(text + "\n" + js + "\n")
