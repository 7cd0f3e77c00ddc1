"""BASELINE configs[1] through the C entry points alone: export the 512x512 4-step ControlNet program as a plan file (one frame and
five frames per launch), compile examples/plan_host.c and let it run the frames -- no Python in the denoising process.
    python scripts/plan_bench.py [--dir /tmp]"""
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import config as C, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402
from videosd_amd.plan import export_plan  # noqa: E402

d = sys.argv[sys.argv.index("--dir") + 1] if "--dir" in sys.argv else "/tmp"
exe = os.path.join(d, "plan_host")
lib = os.path.join(ROOT, "videosd_amd")
subprocess.run(["gcc", "-O2", os.path.join(ROOT, "examples", "plan_host.c"), "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lvsd", "-Wl,-rpath," + lib,
                "-o", exe], check=True)
ops = HipOps(0)
ops.load_tuning(os.environ.get("VSD_TUNING") or os.path.join(ROOT, "profiles", "tuning_mi355x.json"))
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
for B, lanes in ((1, 1), (5, 1), (5, 4), (1, 4)):
    eng.tune_for_lanes = lanes >= 3 and B > 1  # (bench.py's and the drop-in class's rule: coalesced launches on busy lanes take the throughput-mode forms)
    eng.prepare(512, 512, 4, 0.6, use_controlnet=True, batch=B)
    f = np.random.default_rng(0).integers(0, 256, (512, 512, 3) if B == 1 else (B, 512, 512, 3), dtype=np.uint8)
    want = eng.infer_u8(f).copy()
    ts = []
    for _ in range(10):
        t = time.perf_counter()
        eng.infer_u8(f)
        ts.append(time.perf_counter() - t)
    plan = os.path.join(d, f"sd15_512_b{B}.vsdplan")
    t = time.perf_counter()
    info = export_plan(eng, plan)
    t_exp = time.perf_counter() - t
    open(os.path.join(d, "in.raw"), "wb").write(f.tobytes())
    t = time.perf_counter()
    r = subprocess.run([exe, plan, os.path.join(d, "in.raw"), os.path.join(d, "out.raw"), str(30 * lanes), str(lanes)], capture_output=True, text=True, timeout=600)
    t_host = time.perf_counter() - t
    got = np.frombuffer(open(os.path.join(d, "out.raw"), "rb").read(), dtype=np.uint8).reshape(f.shape)
    print(f"frames per launch {B}, {lanes} lane(s): plan {os.path.getsize(plan) / 1e9:.2f} GB ({info['regions']} regions, {info['calls']} calls, scratch {info['scratch_bytes'] / 1e9:.2f} GB), "
          f"export {t_exp:.1f} s; C host (load + {30 * lanes} launches: {t_host:.1f} s): {r.stdout.strip() or r.stderr.strip()[-300:]}; "
          f"Python engine, one-stream form, same frames: {1e3 * float(np.median(ts)):.2f} ms per launch; bit-identical: {bool(np.array_equal(got, want))}", flush=True)
    os.remove(plan)
