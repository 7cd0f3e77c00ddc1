"""Concurrency soak: the bench configuration (SOAK_BATCH frames per launch, SOAK_LANES launches in flight on as many launch lanes;
SOAK_SIDE=1: two lanes, each with its ControlNet encoder on the lane's side stream -- all four launch streams busy)
replayed a few hundred times; every result must be bit-identical to the sequential result of the same input (no
interference between the slots, no state leaking across replays). On a mismatch the first differing stage buffer is
named. Exit code 1 on any mismatch.   python scripts/soak.py [launches]   env: SOAK_BATCH (3), SOAK_SIZE (512), SOAK_NO_CN, SOAK_EAGER"""
import collections, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import config as C, weights as W
from videosd_amd.engine import Engine
from videosd_amd.ops import HipOps


def run(n=300, batch=3, controlnet=True, use_graph=True, size=512, verbose=True, lanes=2, side=False):
    ops = HipOps(0)
    ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
    wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")
    wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")
    wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
    eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds((torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half())
    lanes = 2 if side else max(1, min(4, lanes))
    eng.overlap_controlnet = bool(side)
    eng.overlap_launch = bool(side)
    eng.tune_for_lanes = lanes >= 3 and batch > 1   # (bench.py's and the drop-in class's rule)
    eng.prepare(size, size, 4, 0.6, use_controlnet=controlnet, batch=batch, use_graph=use_graph)
    slots = [eng]
    while len(slots) < lanes:
        sl = eng.make_slot()
        sl.prepare(size, size, 4, 0.6, use_controlnet=controlnet, batch=batch, use_graph=use_graph)
        slots.append(sl)
    rng = np.random.default_rng(0)
    shape = (size, size, 3) if batch == 1 else (batch, size, size, 3)
    inputs = [rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(4)]
    ref = [eng.infer_u8(x).copy() for x in inputs]
    names = [k for k in ("control", "cond_emb", "x0", "eps", "denoised", "dec_in", "dec_out") if k in eng.buffers]
    ref_bufs = []
    for x in inputs:
        eng.infer_u8(x)
        ref_bufs.append({k: eng.buffers[k].clone() for k in names})
    bad, where, stat = 0, collections.Counter(), collections.Counter()

    def check(pe, pk):
        nonlocal bad
        out = pe.collect_u8()
        if not np.array_equal(out, ref[pk]):
            bad += 1
            d = np.abs(out.astype(int) - ref[pk].astype(int))
            stat[(slots.index(pe), pk, int(d.max()))] += 1
            for kname in names:
                if not torch.equal(pe.buffers[kname], ref_bufs[pk][kname]):
                    where[kname] += 1
                    break

    for si, e in enumerate(slots):  # sequential on each slot first
        for k in range(4):
            if not np.array_equal(e.infer_u8(inputs[k]), ref[k]):
                bad += 1
                if verbose: print("sequential mismatch: slot", si, "input", k, flush=True)
    t0 = time.time()
    pending = []
    for i in range(n):
        e = slots[i % lanes]
        if len(pending) == lanes:
            check(*pending.pop(0))
        e.submit_u8(inputs[i % 4])
        pending.append((e, i % 4))
    for pe, pk in pending:
        check(pe, pk)
    dt = time.time() - t0
    if verbose:
        print("first differing stage buffer (count over mismatching launches):", dict(where))
        for k, v in sorted(stat.items()):
            print("mismatch (slot, input, max|diff|):", k, "x", v)
        print(f"soak: {n} launches x {batch} frames in {dt:.1f} s ({batch * n / dt:.1f} frames/s incl. host copies and compares), mismatches: {bad}")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    sys.exit(1 if run(n, int(os.environ.get("SOAK_BATCH", "3")), not os.environ.get("SOAK_NO_CN"), not os.environ.get("SOAK_EAGER"),
                      size=int(os.environ.get("SOAK_SIZE", "512")), lanes=int(os.environ.get("SOAK_LANES", "2")),
                      side=bool(os.environ.get("SOAK_SIDE"))) else 0)
