"""In-situ tuning of the throughput-mode conv forms (round 6): coordinate descent on the frames/s of the TIMED program itself.

The throughput-mode table (profiles/tuning_mi355x.json, key's last field 1) is filled by timing every candidate of a shape with four
copies of ITSELF in flight (ops.tune_conv) -- a proxy for a shared chip that round 5 showed is good to a few per cent, not to one: a
form that wins by 3 % against three copies of itself is not a form that wins beside the other layers.  This script asks the question
that counts: with the whole B x 4 program replaying on the four launch lanes, does the frame rate go up when layer shape X runs in
form Y?  For the shapes that carry the most time it tries the proxy's next-best forms one at a time (all four lanes' graphs captured
again, frames/s of a few dozen launches), keeps a change only if it is confirmed above the noise, and writes the table.

    python scripts/tune_in_situ.py [--batch 5] [--shapes 40] [--alts 3] [--seconds 1500] [out.json]

--batch 1: the ONE-frame program (latency-mode entries, key's last field 0) on four busy lanes, with a guard: a change must also leave
the LONE frame's latency (two-stream form, nothing else on the GPU) where it was (--guard-ms, default 0.05) -- forms that cost a
busy chip less without costing the lone frame anything.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import config as C, weights as W  # noqa: E402
from videosd_amd.engine import Engine  # noqa: E402
from videosd_amd.ops import HipOps  # noqa: E402


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


B, NSHAPES, NALTS, BUDGET = arg("--batch", 5), arg("--shapes", 40), arg("--alts", 3), arg("--seconds", 1500.0)
GAIN = arg("--gain", 1.004)  # a change must beat the incumbent by this factor, twice
GUARD_MS = arg("--guard-ms", 0.05)
LANES = B > 1  # (bench.py's and the drop-in class's rule: coalesced launches on three or four lanes run the throughput-mode forms)
path = os.environ.get("VSD_TUNING") or os.path.join(ROOT, "profiles", "tuning_mi355x.json")
out = next((a for a in sys.argv[1:] if a.endswith(".json")), path)
S = 4

ops = HipOps(0)
ops.load_tuning(path)
wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
pool = [eng] + [eng.make_slot() for _ in range(S - 1)]
frame = np.random.default_rng(0).integers(0, 256, (B, 512, 512, 3) if B > 1 else (512, 512, 3), dtype=np.uint8)


def prepare_all():
    for e in pool:
        e.overlap_controlnet = True
        e.overlap_launch = False
        e.tune_for_lanes = LANES
        e.prepare(512, 512, 4, 0.6, use_controlnet=True, batch=B)
    for e in pool:
        e.infer_u8(frame)


def fps(reps=3):
    res = []
    n = max(4 * S, 80 // B)
    for _ in range(reps):
        for e in pool:
            e.ops.synchronize()
        t = time.perf_counter()
        for i in range(n):
            pool[i % S].launch()
        for e in pool:
            e.ops.synchronize()
        res.append(n * B / (time.perf_counter() - t))
    return float(np.median(res))


def lone_ms(n=24):
    """p50 of a lone frame on lane 0 (two-stream form), nothing else in flight"""
    e = pool[0]
    for o in pool:
        o.ops.synchronize()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        e.launch(overlap=True)
        e.ops.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


t_all = time.time()
prepare_all()
base = fps(5)
print(f"{B} x {S}: {base:.2f} frames/s with the table as loaded ({len(ops.tile_override)} entries)", flush=True)

# the program's conv shapes (one-stream form: what four lanes replay), their launch counts and the proxy's candidate tables
ops.tune_mode = 1 if LANES else 0
shapes = {}
for fn, a, k in Engine.flat_calls(eng.program_serial.calls):
    if fn.__name__ != "conv" or k.get("tile") is not None:
        continue
    key = ops.conv_key_of(a[2], a[3], k)
    s = shapes.setdefault(key, dict(n=0, a=a, k=k))
    s["n"] += 1
print(f"{len(shapes)} conv shapes in the program; timing their candidates (four copies in flight) ...", flush=True)
for key, s in shapes.items():
    cur = ops.tile_override.get(key)
    try:
        _, table = ops.tune_conv(s["a"], s["k"])
    except RuntimeError:
        table = []
    finally:
        if cur is not None:
            ops.tile_override[key] = cur
        else:
            ops.tile_override.pop(key, None)
    s["table"] = table
    s["cur"] = cur
    cur_us = next((t[0] for t in table if cur is not None and tuple(t[1:]) == tuple(cur)), table[0][0] if table else 0.0)
    s["weight"] = s["n"] * cur_us
order = sorted(shapes, key=lambda kk: -shapes[kk]["weight"])[:NSHAPES]
print(f"candidates timed ({time.time() - t_all:.0f} s); trying the {len(order)} heaviest shapes in the running program", flush=True)

best = fps(5)
lone0 = lone_ms() if not LANES else 0.0
if not LANES:
    print(f"lone frame {lone0:.3f} ms", flush=True)
changed = []
for key in order:
    if time.time() - t_all > BUDGET:
        print("time budget spent", flush=True)
        break
    s = shapes[key]
    cur = s["cur"]
    alts = [tuple(t[1:]) for t in s["table"] if cur is None or tuple(t[1:]) != tuple(cur)][:NALTS]
    kept = cur
    for alt in alts:
        ops.tile_override[key] = (int(alt[0]), int(alt[1]), bool(alt[2]), int(alt[3]))
        try:
            prepare_all()
            f1 = fps(3)
            ok = f1 > best * GAIN and fps(3) > best * GAIN
            if ok and not LANES:
                l1 = lone_ms()
                ok = l1 <= lone0 + GUARD_MS
                print(f"      lone frame {l1:.3f} ms against {lone0:.3f}", flush=True)
        except RuntimeError as e:
            print("   ", key[:4], alt, "refused:", str(e)[:80], flush=True)
            f1, ok = 0.0, False
        print(f"  M={key[0]} N={key[1]} K={key[2]} ks={key[3]} x{s['n']}: {kept} -> {alt}: {f1:.2f} against {best:.2f}" + ("  KEPT" if ok else ""), flush=True)
        if ok:
            best = max(f1, best * GAIN)
            kept = alt
            changed.append((key, cur, alt, f1))
        else:
            if kept is not None:
                ops.tile_override[key] = (int(kept[0]), int(kept[1]), bool(kept[2]), int(kept[3]))
            else:
                ops.tile_override.pop(key, None)
prepare_all()
final = fps(5)
print(f"{B} x {S}: {base:.2f} -> {final:.2f} frames/s, {len(changed)} entries changed ({time.time() - t_all:.0f} s)")
for key, cur, alt, f in changed:
    print("  ", list(key), cur, "->", alt, f"{f:.2f}")
old = {}
if os.path.exists(path):
    for k, v in json.load(open(path)).get("table", []):
        old[tuple(k)] = tuple(v)
for key, cur, alt, f in changed:
    old[tuple(key)] = (int(alt[0]), int(alt[1]), bool(alt[2]), int(alt[3]))
json.dump({"device": torch.cuda.get_device_name(ops.device), "table": [[list(k), list(v)] for k, v in sorted(old.items(), key=str)]},
          open(out, "w"), indent=0)
print(f"{len(old)} entries -> {out}")
