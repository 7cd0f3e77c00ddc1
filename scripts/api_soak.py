"""Long run through the drop-in class in a worker process: sessions that change prompt, frame size, step count and sliders
while frames stream (more programs than `max_plans`, more prompts than the prompt LRU: plans and prompt blocks are evicted and
rebuilt all the time).  Checks, per phase: every frame comes back; a frame seen before under the same (session, prompt) is the
same picture again (mean |diff| < 0.5 LSB: the worker coalesces a varying number of frames per launch, whose kernels differ in
rounding, so not bit-equal); no picture is flat; device memory does not drift; throughput does not decay.  Exit code 1 on any.

usage (GPU box): python scripts/api_soak.py [seconds=120] [lanes=4] [batch=5] [memory_budget=0.6]
(a small memory_budget, e.g. 0.15, makes the plan cache evict engines for memory all the time)"""
import asyncio, json, os, sys, time
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from videosd_amd.pipeline import VideoSDPipeline

SESSIONS = [
    dict(height=512, width=512, strength=0.6, steps=4, controlnet_scale=1.0),
    dict(height=384, width=640, strength=0.5, steps=4, controlnet_scale=1.2),
    dict(height=256, width=256, strength=0.6, steps=2, controlnet_scale=0.8),
    dict(height=512, width=512, strength=0.9, steps=4, controlnet_scale=1.0),   # another timestep count: another program
    dict(height=512, width=512, strength=0.6, steps=4, controlnet_scale=0.5),   # a slider step of session 0: same program
]
PROMPTS = ["pixar, cg", "an oil painting of a harbour at dusk", "lego bricks", "watercolour, autumn", "neon city at night", "charcoal sketch"]


def main(seconds=120.0, lanes=4, batch=5, memory_budget=0.6):
    frames = [Image.fromarray(f, "RGB") for f in bench.synthetic_frames(8, 512, 512)]
    w = VideoSDPipeline.remote(model="SimianLuo/LCM_Dreamshaper_v7", controlnet="lllyasviel/control_v11p_sd15_canny", device=0,
                               batch=batch, lanes=lanes, shm_slots=(lanes + 1) * batch + 4, call_timeout=600.0, memory_budget=memory_budget)
    seen, bad, log = {}, [], []
    rng = np.random.default_rng(5)

    def mem(key="allocated_mb"):
        return w.metrics().get("pipeline", {}).get(key)

    async def phase(si, pi, n, depth):
        opts = dict(SESSIONS[si], prompt=PROMPTS[pi], seed=23)
        sem = asyncio.Semaphore(depth)
        outs = [None] * n

        async def one(i):
            async with sem:
                outs[i] = await w.infer.remote(frames[i % len(frames)], **opts)

        t0 = time.perf_counter()
        await asyncio.gather(*[one(i) for i in range(n)])
        dt = time.perf_counter() - t0
        for i, im in enumerate(outs):
            if im is None:
                bad.append(("missing", si, pi, i))
                continue
            a = np.asarray(im)
            if a.shape != (opts["height"], opts["width"], 3):
                bad.append(("shape", si, pi, i, a.shape))
                continue
            if int(a.max()) - int(a.min()) < 8:
                bad.append(("flat", si, pi, i))
            k = (si, pi, i % len(frames))
            if k in seen:
                d = float(np.abs(a.astype(np.int16) - seen[k].astype(np.int16)).mean())
                if d >= 0.5:
                    bad.append(("changed", si, pi, i, round(d, 3)))
            elif len(seen) < 400:
                seen[k] = a
        return n / dt

    try:
        w.infer(frames[0], prompt=PROMPTS[0], seed=23, **SESSIONS[0])
        m0, t_end, ph, samples = mem(), time.time() + seconds, 0, []
        first_fps = {}
        while time.time() < t_end:
            si, pi = int(rng.integers(len(SESSIONS))), int(rng.integers(len(PROMPTS)))
            n = int(rng.choice([1, 3, 7, 24, 60]))
            depth = int(rng.choice([1, batch, (lanes + 1) * batch]))
            fps = asyncio.run(phase(si, pi, n, depth))
            ph += 1
            if n >= 24 and depth > batch:
                first_fps.setdefault(si, []).append(round(fps, 1))
            log.append((ph, si, pi, n, depth, round(fps, 1)))
            if ph % 5 == 0:
                samples.append(mem())
            if ph % 20 == 0:
                print(f"phase {ph}: session {si} prompt {pi} n {n} depth {depth}: {fps:.1f} frames/s; allocated {samples[-1]} MB, free {mem('device_free_mb')} MB; bad {len(bad)}", flush=True)
        m1 = mem()
        met = w.metrics()
    finally:
        w.close()
    # memory: programs / prompt blocks are cached up to their LRU sizes (3 programs x up to 5 batch sizes x 4 lanes of arenas),
    # so the first phases grow it; a leak keeps growing: the last third must stay under what the first two thirds reached
    k = (2 * len(samples)) // 3
    early, late = (max(samples[:k]), max(samples[k:])) if k and samples[k:] else (None, None)
    decay = {si: (v[0], v[-1]) for si, v in first_fps.items() if len(v) >= 2}
    print(json.dumps({"phases": ph, "frames": sum(r[3] for r in log), "bad": bad[:10], "n_bad": len(bad), "allocated_mb_start": m0, "allocated_mb_end": m1, "allocated_mb_max_first_two_thirds": early, "allocated_mb_max_last_third": late,
                      "fps_first_last_by_session": decay, "worker": {k: met.get(k) for k in ("launches", "frames", "frames_per_launch")},
                      "stage_ms_p50": met.get("pipeline", {}).get("stage_ms_p50")}))
    leak = early is not None and late > 1.05 * early + 1024
    evicted = (met.get("pipeline", {}).get("stage_ms_p50") or {}).get("engines_evicted_for_memory", 0)
    # (with a budget small enough to evict all the time a phase's rate is set by how many engines it has to prepare again: no
    #  decay criterion then -- results and the memory bound are what that run checks)
    slow = not evicted and any(b < 0.8 * a for a, b in decay.values())
    if bad or leak or slow:
        print("FAILED:", "results" if bad else "", "memory" if leak else "", "throughput decay" if slow else "")
        return 1
    print("api soak passed")
    return 0


if __name__ == "__main__":
    a = sys.argv[1:]
    sys.exit(main(float(a[0]) if a else 120.0, int(a[1]) if len(a) > 1 else 4, int(a[2]) if len(a) > 2 else 5, float(a[3]) if len(a) > 3 else 0.6))
