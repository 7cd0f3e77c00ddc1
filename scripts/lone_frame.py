"""One frame per launch (BASELINE configs[1] as worded): latency of a lone frame and throughput of one-frame launches on 3 / 4
lanes, with the launch count of the one-frame program split by kind.

    python scripts/lone_frame.py [--retune] [--save-tuning PATH] [--no-cn] [--tag NAME] [--lanes]

--retune         time every conv shape of the one-frame program again (the table's entries for them are ignored)
--save-tuning    merge this run's choices into the table at PATH
--lanes          also 1 x 3 and 1 x 4 (one-frame launches on three / four lanes)
--mode1          the one-frame program in THROUGHPUT-mode kernel forms (every candidate timed with four lanes busy, online: minutes) --
                 what one-frame launches would cost on busy lanes if they were tuned like the coalesced ones
Prints one JSON line (and appends it to gpurun_out/lone_frame.jsonl)."""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import config as C, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402

argv = sys.argv[1:]
retune = "--retune" in argv
use_cn = "--no-cn" not in argv
tag = argv[argv.index("--tag") + 1] if "--tag" in argv else ""
save = argv[argv.index("--save-tuning") + 1] if "--save-tuning" in argv else None
table = os.environ.get("VSD_TUNING") or os.path.join(ROOT, "profiles", "tuning_mi355x.json")

ops = HipOps(0)
n_loaded = ops.load_tuning(table)
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
eng.overlap_launch = True
mode1 = "--mode1" in argv
eng.tune_for_lanes = mode1
ops.tune_lanes_online = mode1
if retune:
    # record the one-frame program once without tuning to learn its conv keys, drop them from the table, prepare again
    eng.prepare(512, 512, 4, 0.6, use_controlnet=use_cn, batch=1, autotune=False, use_graph=False)
    drop = set()
    for fn, a, k in eng.program.calls:
        if fn.__name__ == "conv":
            drop.add(ops.conv_key_of(a[2], a[3], k))
    for key in drop:
        ops.tile_override.pop(key, None)
t0 = time.perf_counter()
plan = eng.prepare(512, 512, 4, 0.6, use_controlnet=use_cn, batch=1, use_graph=True)
prepare_s = time.perf_counter() - t0

launches, kinds = eng.launches_by_kind()  # what a graph replay issues: the two-stream form (a lone launch) ...
launches_serial, kinds_serial = eng.launches_by_kind(serial=True)  # ... and the one-stream form (a launch among busy lanes)

frames = np.random.default_rng(0).integers(0, 256, (12, 512, 512, 3), dtype=np.uint8)
for i in range(5):
    eng.infer_u8(frames[i])
lat, gpu = [], []
for i in range(40):
    t1 = time.perf_counter()
    eng.infer_u8(frames[i % 12])
    lat.append((time.perf_counter() - t1) * 1e3)
    gpu.append(getattr(eng, "last_gpu_ms", 0.0))
out = {"tag": tag, "controlnet": use_cn, "retune": retune, "table_entries_loaded": n_loaded, "prepare_s": round(prepare_s, 2),
       "p50_ms": round(statistics.median(lat), 3), "min_ms": round(min(lat), 3), "gpu_p50_ms": round(statistics.median(gpu), 3),
       "n_ops": plan["n_ops"], "launches": launches, "launches_by_kind": kinds, "launches_one_stream_form": launches_serial,
       "launches_by_kind_one_stream_form": kinds_serial, "graphs": plan.get("graphs"), "edges": plan.get("edges")}

# the serial sequence (everything on the lane's own stream): what a launch takes when three or four lanes are busy
ops.synchronize()
t1 = time.perf_counter()
for i in range(20):
    eng.launch(overlap=False)
ops.synchronize()
out["serial_ms"] = round((time.perf_counter() - t1) / 20 * 1e3, 3)

if "--lanes" in argv:
    pool = [eng]
    for s in (3, 4):
        while len(pool) < s:
            sl = eng.make_slot()
            sl.overlap_launch = False
            sl.tune_for_lanes = mode1
            sl.prepare(512, 512, 4, 0.6, use_controlnet=use_cn, batch=1)
            sl.infer_u8(frames[0])
            pool.append(sl)
        best = 0.0
        for _rep in range(3):
            for e in pool:
                e.ops.synchronize()
            n = 40 * s // 4
            t1 = time.perf_counter()
            for i in range(n):
                pool[i % s].launch(overlap=False)
            for e in pool:
                e.ops.synchronize()
            best = max(best, n / (time.perf_counter() - t1))
        out[f"fps_1x{s}"] = round(best, 2)

if save:
    ops.save_tuning(save)
line = json.dumps(out)
print(line, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "lone_frame.jsonl"), "a") as f:
    f.write(line + "\n")
