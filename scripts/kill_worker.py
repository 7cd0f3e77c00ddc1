"""A real GPU worker killed mid-stream (SIGKILL): the frames it held fail with WorkerDied, the dispatcher starts a fresh process
(new HIP context, new engine, warmed up) and the stream continues with the same pictures as before.
usage (GPU box): python scripts/kill_worker.py"""
import asyncio, os, signal, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
import bench
from videosd_amd.dispatch import FrameDispatcher, WorkerDied
from videosd_amd.pipeline import VideoSDPipeline

opts = dict(prompt="pixar, cg", height=256, width=256, strength=0.6, steps=2, controlnet_scale=1.0, seed=23)


async def go(w, frames, holder):
    d = holder["d"] = FrameDispatcher([w], respawn=True, depth=6, warm_options=dict(opts))
    first_pid = d.pipelines[0]._proc.pid
    want = {}
    for i in range(6):  # reference pictures, one at a time
        d.submit(frames[i], **opts)
        tk, res = await asyncio.wait_for(d.next_result(), timeout=120)
        want[i] = np.asarray(res)
    sent = [d.submit(frames[i % 6], **opts) for i in range(6)]
    os.kill(first_pid, signal.SIGKILL)
    died = ok = 0
    for _ in sent:
        tk, res = await asyncio.wait_for(d.next_result(), timeout=120)
        if isinstance(res, WorkerDied):
            died += 1
        else:
            ok += 1
    print(f"after the kill: {died} frames failed with WorkerDied, {ok} had already come back", flush=True)
    assert died >= 1
    t0 = time.time()
    while not d.healthy[0]:
        assert time.time() - t0 < 180, "no replacement worker"
        await asyncio.sleep(0.1)
    print(f"replacement worker pid {d.pipelines[0]._proc.pid} (was {first_pid}) ready after {time.time() - t0:.1f} s; respawns {d.respawns}", flush=True)
    assert d.pipelines[0]._proc.pid != first_pid and d.respawns == 1
    bad = 0
    for rnd in range(4):
        sent = [d.submit(frames[i], **opts) for i in range(6)]
        got = {}
        for _ in sent:
            tk, res = await asyncio.wait_for(d.next_result(), timeout=120)
            assert not isinstance(res, Exception), res
            got[tk] = np.asarray(res)
        for j, tk in enumerate(sent):
            if np.abs(got[tk].astype(int) - want[j].astype(int)).mean() >= 0.5:
                bad += 1
    assert bad == 0, f"{bad} frames of the new worker differ from the old worker's pictures"
    print("kill_worker passed: 24 frames through the replacement, same pictures")

if __name__ == "__main__":  # (the worker is a spawned child: it imports this module again)
    frames = [Image.fromarray(f, "RGB") for f in bench.synthetic_frames(6, 256, 256)]
    w = VideoSDPipeline.remote(model="SimianLuo/LCM_Dreamshaper_v7", controlnet="lllyasviel/control_v11p_sd15_canny", device=0, batch=3,
                               lanes=2, call_timeout=120.0, tuning_mode="table")
    holder = {}
    try:
        asyncio.run(go(w, frames, holder))
    finally:
        for p in [w] + (list(holder["d"].pipelines) if "d" in holder else []):  # (the replacement too: it owns new shared memory)
            try:
                p.close()
            except Exception:
                pass
