"""Host-side ceiling of the N-worker product path (reference: diffusert/server.py:104-143, 317-321).

One asyncio parent -- `FrameDispatcher` over N `RemotePipeline` workers -- moves every frame of the node: PIL 512x512 in ->
shared-memory request slot -> worker process -> reply slot -> PIL out.  At 8 x 137 frames/s the parent has 0.91 ms per frame.
Here the workers are zero-cost stand-ins (tests/helpers_fake_pipeline.py NullPipeline: no GPU, no pixel work), so what is
measured is the transport and the parent: frames/s, and the parent's own milliseconds per frame by stage.  No GPU needed.

    python scripts/dispatch_ceiling.py [--workers 8] [--frames 4000] [--depth 20] [--batch 5] [--lanes 4] [--size 512]
Prints one JSON line."""
import asyncio
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")


def measure(workers=8, frames=4000, depth=20, batch=5, lanes=4, size=512, mode="in_order"):
    import numpy as np
    from PIL import Image

    from videosd_amd.dispatch import FrameDispatcher, RemotePipeline

    ws = [RemotePipeline(factory="helpers_fake_pipeline:NullPipeline", model="m", controlnet="c", batch=batch, lanes=lanes,
                         shm_slots=depth + 4, shm_slot_bytes=size * size * 3, wait=False) for _ in range(workers)]
    try:
        for w in ws:
            w.wait_ready()
        rng = np.random.default_rng(0)
        imgs = [Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8), "RGB") for _ in range(8)]
        opts = dict(prompt="pixar, cg", height=size, width=size, strength=0.6, steps=4)

        async def run(n):
            d = FrameDispatcher(ws, mode=mode, depth=depth)
            got = sub = 0
            t_submit = t_result = 0.0
            while got < n:
                # keep every worker's queue full (the ceiling, not a paced stream), then take what has finished
                while sub < n and d.pending < workers * depth:
                    t0 = time.perf_counter()
                    t = d.submit(imgs[sub % 8], **opts)
                    t_submit += time.perf_counter() - t0
                    if t is None:
                        break
                    sub += 1
                t0 = time.perf_counter()
                _t, res = await d.next_result()
                t_result += time.perf_counter() - t0
                assert not isinstance(res, Exception), res
                assert res.size == (size, size)
                got += 1
            return t_submit, d.dropped

        asyncio.run(run(min(200, frames)))  # warm-up: imports, first slots, the workers' cached reply image
        for w in ws:
            for k in w.host_s:
                w.host_s[k] = 0 if k == "frames" else 0.0
        cpu0, t0 = time.process_time(), time.perf_counter()
        t_submit, dropped = asyncio.run(run(frames))
        wall, cpu = time.perf_counter() - t0, time.process_time() - cpu0
        stage = {k: round(sum(w.host_s[k] for w in ws) / frames * 1e3, 4) for k in ("slot_write", "send", "slot_read", "complete")}
        stage["submit_total"] = round(t_submit / frames * 1e3, 4)  # (slot write + pickle header + queue + coroutine start)
        return {"workers": workers, "frames": frames, "depth_per_worker": depth, "batch": batch, "lanes": lanes, "frame": f"{size}x{size} RGB",
                "fps": round(frames / wall, 1), "parent_cpu_ms_per_frame": round(cpu / frames * 1e3, 4),
                "parent_wall_ms_per_frame": round(wall / frames * 1e3, 4), "parent_stage_ms_per_frame": stage,
                "budget_ms_per_frame_at_8x137fps": 0.91, "host_cores": os.cpu_count(), "dropped": dropped, "mode": mode}
    finally:
        for w in ws:
            w.close()


if __name__ == "__main__":
    a = sys.argv[1:]

    def arg(name, default):
        return int(a[a.index(name) + 1]) if name in a else default

    if "--switch-interval-us" in a:  # (experiment: the interpreter's GIL hand-over interval, default 5000 us)
        sys.setswitchinterval(arg("--switch-interval-us", 5000) * 1e-6)
    print(json.dumps(measure(arg("--workers", 8), arg("--frames", 4000), arg("--depth", 20), arg("--batch", 5), arg("--lanes", 4),
                             arg("--size", 512))), flush=True)
