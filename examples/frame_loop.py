#!/usr/bin/env python3
"""The reference's frame loop without WebRTC: a synthetic camera feeds `VideoSDPipeline.remote(...)` workers exactly the
way diffusert/server.py does (`VideoSDTrack.recv` -> `diffuse` -> `await pipelines[gpu].infer.remote(img, **options)`,
server.py:104-143), one worker process per GPU, frames sharded round-robin, drop-if-busy, newest-wins or in-order display.

  python examples/frame_loop.py --gpus 1 --fps 60 --seconds 10 --batch 3          (needs the MI355X)
  python examples/frame_loop.py --factory helpers_fake_pipeline:FakePipeline ...   (any stand-in with the same surface)

Prints one JSON line: frames offered / processed / dropped, output frames per second, p50 / p95 submit->result latency.
"""
import argparse
import asyncio
import json
import os
import statistics
import sys
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd.dispatch import FrameDispatcher, RemotePipeline  # noqa: E402


def camera_frame(k: int, w: int, h: int) -> Image.Image:
    """A moving gradient with seeded noise (Sobel maximum never 0: canny_gpu.py:39 would divide by zero)."""
    rng = np.random.default_rng(k)
    yy, xx = np.mgrid[0:h, 0:w]
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) // 2 + ((xx * 2 + yy + 17 * k) % 256).astype(np.uint8)[..., None] // 2
    return Image.fromarray(a.astype(np.uint8), "RGB")


async def run(args):
    cfg = dict(model=args.model, controlnet=args.controlnet)
    extra = json.loads(args.worker_config) if args.worker_config else {}
    workers = [RemotePipeline(factory=args.factory, batch=args.batch, device=i, **cfg, **extra) for i in range(args.gpus)]
    opts = dict(prompt=args.prompt, height=args.height, width=args.width, strength=args.strength, steps=args.steps,
                controlnet_scale=args.controlnet_scale, seed=23)
    try:
        if not args.no_warmup:  # the first call of a worker builds and captures its graph (compile_model in the reference)
            await asyncio.gather(*[w.infer.remote(camera_frame(0, 640, 480), **opts) for w in workers])
        disp = FrameDispatcher(workers, mode=args.mode, depth=args.depth or args.batch * 2)
        pool = [camera_frame(i, args.cam_width, args.cam_height) for i in range(16)]  # the camera itself costs nothing
        t_sub, lat, shown = {}, [], 0
        period = 1.0 / args.fps
        t0 = time.perf_counter()

        async def display():
            nonlocal shown
            while True:
                ticket, img = await disp.next_result()
                if isinstance(img, Exception):
                    raise img
                lat.append((time.perf_counter() - t_sub[ticket]) * 1e3)
                shown += 1

        shower = asyncio.ensure_future(display())
        k = 0
        while time.perf_counter() - t0 < args.seconds:
            frame = pool[k % len(pool)]
            now = time.perf_counter()
            ticket = disp.submit(frame, **opts)
            if ticket is not None:
                t_sub[ticket] = now
            k += 1
            await asyncio.sleep(max(0.0, t0 + k * period - time.perf_counter()))
        while disp.pending:
            await asyncio.sleep(0.01)
        wall = time.perf_counter() - t0
        shower.cancel()
        out = {"offered": k, "processed": disp.submitted, "dropped": disp.dropped, "shown": shown, "seconds": round(wall, 2),
               "output_fps": round(shown / wall, 2), "p50_latency_ms": round(statistics.median(lat), 1) if lat else None,
               "p95_latency_ms": round(sorted(lat)[int(0.95 * (len(lat) - 1))], 1) if lat else None,
               "config": {"gpus": args.gpus, "camera_fps": args.fps, "batch": args.batch, "mode": args.mode, **opts}}
        print(json.dumps(out), flush=True)
        return out
    finally:
        for w in workers:
            w.close()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--fps", type=float, default=60.0, help="camera frame rate")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--batch", type=int, default=3, help="frames a worker may coalesce into one launch")
    ap.add_argument("--depth", type=int, default=0, help="frames one worker may hold (default 2 x batch)")
    ap.add_argument("--mode", default="latest", choices=["latest", "in_order"])
    ap.add_argument("--factory", default="videosd_amd.pipeline:VideoSDPipeline")
    ap.add_argument("--worker-config", default="", help="JSON of extra worker kwargs")
    ap.add_argument("--model", default="SimianLuo/LCM_Dreamshaper_v7")
    ap.add_argument("--controlnet", default="lllyasviel/control_v11p_sd15_canny")
    ap.add_argument("--prompt", default="pixar, cg")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--strength", type=float, default=0.6)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--controlnet-scale", dest="controlnet_scale", type=float, default=1.0)
    ap.add_argument("--cam-width", dest="cam_width", type=int, default=640)
    ap.add_argument("--cam-height", dest="cam_height", type=int, default=480)
    ap.add_argument("--no-warmup", action="store_true")
    return asyncio.run(run(ap.parse_args(argv)))


if __name__ == "__main__":
    main()
