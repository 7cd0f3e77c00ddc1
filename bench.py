#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the per-frame SD1.5 LCM denoising path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one frame through the whole hot path (BASELINE.json config 2, reference-faithful: u8 512x512
frame resident in HBM -> Sobel/ControlNet conditioning -> TAESD encode -> 4 x (ControlNet + UNet + LCM step)
-> TAESD decode -> u8 frame in HBM).  Frames are independent (the reference resets its RNG per frame), so the engine
takes `--batch` of them per hipGraph replay (stacked along the GEMM M dimension: one pass over the 2.45 GB of weights
serves all of them; every frame keeps its own GroupNorm statistics / attention / Sobel maximum) and keeps `--slots`
launches in flight, one per launch lane (a lane = one of the process's four launch streams: four hardware queues on four
command-processor pipes); K frames = ceil(K / batch) launches.  `--batch 1 --slots 1` is one frame at a time; the
single-frame latency (`p50_latency_ms`) is always measured that way.

N GPUs: one process per GPU.  `--gpus N` without WORLD_SIZE in the environment makes THIS process a launcher that never
touches the GPU: it starts N fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one rank per
GPU), waits for them and relays rank 0's JSON line; under `python -m torch.distributed.run ... bench.py --gpus N` (the
driver's form) the environment is already there and this process is a rank.  Frames are sharded round-robin
(frame k -> rank k mod N, no data-path collective; the prompt embeddings are broadcast once from rank 0 over RCCL), every
rank does K frames (weak scaling) and value = N*K / max-over-ranks time.

Weights are seeded synthetic tensors of the SD1.5 / ControlNet / TAESD architectures (no network for
checkpoints), inputs are synthetic frames: data = "synthetic".
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
H = W = 512
LCM_STEPS = 4
STRENGTH = 0.6
METRIC = "frames/sec (whole node) + p50 per-frame latency, SD1.5 512x512 LCM 4-step"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--slots", type=int, default=4, help="launches in flight per GPU (independent frames, one launch lane each: the four "
                    "launch streams of a process are four hardware queues on four command-processor pipes; with <= 2 lanes every "
                    "lane also runs its ControlNet encoder on a side stream)")
    ap.add_argument("--batch", type=int, default=5, help="frames per launch (stacked along the GEMM M dimension); "
                    "--batch 1 --slots 3 is the one-frame-per-launch configuration of the first bench lines")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the leg through VideoSDPipeline.remote(...).infer.remote")
    ap.add_argument("--no-extras", action="store_true", help="headline, latency and roofline only")
    ap.add_argument("--retune", action="store_true", help="ignore profiles/tuning_mi355x.json and time all kernel configs again")
    ap.add_argument("--save-tuning", action="store_true", help="write the tuning table back to profiles/tuning_mi355x.json")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: walk the launch / rendezvous / broadcast / timing "
                    "protocol with a sleep as the step (CPU test of the multi-rank plumbing; the line says dry_run)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ launcher (no GPU here)
def launch_ranks(args, argv):
    """Start `--gpus` ranks of this script as fresh child processes and relay rank 0's line.  This process must not
    initialise HIP (a process that has cannot be replaced or forked safely; the children are separate interpreters)."""
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    line = None
    rcs = [None] * n
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                        "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        # rank 0's stdout is drained by a thread (a full pipe must never block it) while ALL children are polled: the first
        # rank that exits non-zero ends the run at once -- the others would otherwise sit in the rendezvous / a collective
        # until the process-group timeout before anything is reported
        import threading

        out0 = []
        reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout), daemon=True)
        reader.start()
        failed = None
        while any(rc is None for rc in rcs) and failed is None:
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = p.poll()
                    if rcs[i] not in (None, 0):
                        failed = i
            time.sleep(0.05)
        reader.join(timeout=5)
        for ln in out0:
            if ln.startswith("{") and '"metric"' in ln:
                line = ln.strip()
        if failed is not None or line is None:
            sys.stderr.write("".join(out0))
            raise SystemExit(f"bench.py: rank {failed} exited with {rcs[failed]}; the other ranks were stopped" if failed is not None
                             else f"bench.py: ranks exited with {rcs} and rank 0 printed no result line")
    finally:  # (also on KeyboardInterrupt / an exception above): never leave rank processes behind -- exactly these PIDs
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    print(line, flush=True)


# ------------------------------------------------------------------------------------------ rank
def synthetic_frames(n, h, w):
    """seeded noise blended 50% with a moving gradient (Sobel max != 0)."""
    import numpy as np

    rng = np.random.default_rng(1234)
    base = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.empty_like(base)
    for i in range(n):
        grad = ((xx * 2 + yy + 17 * i) % 256).astype(np.uint8)[..., None]
        out[i] = base[i] // 2 + grad // 2
    return out


def build_engine(device_id):
    from videosd_amd import config as C
    from videosd_amd import weights as Wt
    from videosd_amd.engine import Engine
    from videosd_amd.ops import HipOps

    dev = f"cuda:{device_id}"
    ops = HipOps(device_id)
    wu = Wt.synthesize(Wt.unet_spec(C.SD15_UNET), "unet.", device=dev)
    wc = Wt.synthesize(Wt.controlnet_spec(C.SD15_CONTROLNET), "cn.", device=dev)
    wv = Wt.synthesize(Wt.taesd_spec(C.TAESD), "vae.", device=dev)
    eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
    return eng, ops, (wu, wc, wv)


def cpu_baseline(weights, text, frame, budget_s=20.0):
    """The oracle (CPU fp32 restatement of the reference algorithm) on the host cores: BASELINE.md section 3's protocol -- config 1
    (256x256, 1 LCM step: the reference's own CPU-runnable case) 1 warm-up + 3 timed frames, config 2 (512x512, 4 steps: the
    benchmark's workload) 1 timed frame (more while they fit the budget); seconds per frame with the spread.  Returns the baseline
    record and the oracle's config-2 frame (the parity stamp compares the engine's output with it)."""
    import numpy as np
    import torch
    from PIL import Image

    from oracle.pipeline import OraclePipeline
    from videosd_amd import config as C

    wu, wc, wv = ({k: v.float().cpu() for k, v in w.items()} for w in weights)
    orc = OraclePipeline(C.SD15_UNET, C.SD15_CONTROLNET, wu, wc, wv)
    img = Image.fromarray(frame, "RGB")
    t1 = []
    for i in range(4):  # config 1: 1 warm-up + 3 timed
        t0 = time.time()
        orc.infer(img, text[None].float(), height=256, width=256, strength=STRENGTH, steps=1, seed=23, controlnet_scale=1.0, use_controlnet=True)
        if i:
            t1.append(time.time() - t0)
    times = []
    t_all = time.time()
    ref = None
    while len(times) < 3 and (not times or (time.time() - t_all) + max(times) < budget_s):
        t0 = time.time()
        ref = orc.infer(img, text[None].float(), height=H, width=W, strength=STRENGTH, steps=LCM_STEPS, seed=23,
                        controlnet_scale=1.0, use_controlnet=True)
        times.append(time.time() - t0)
    best = min(times)
    rec = {"value": 1.0 / best, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{len(times)} frame(s) of the same 512x512 4-step ControlNet workload: {best:.2f} s/frame (min {min(times):.2f}, max {max(times):.2f}); "
                     f"config 1 (256x256, 1 step, ControlNet): 1 warm-up + 3 frames, {statistics.median(t1):.2f} s/frame (min {min(t1):.2f}, max {max(t1):.2f})",
           "config1_s_per_frame": {"median": round(statistics.median(t1), 3), "min": round(min(t1), 3), "max": round(max(t1), 3), "frames": len(t1)},
           "config2_s_per_frame": {"min": round(min(times), 3), "max": round(max(times), 3), "frames": len(times)}}
    return rec, np.asarray(ref)


def image_parity(got, ref):
    import numpy as np

    d = got.astype(np.float64) - ref.astype(np.float64)
    mse = float((d * d).mean())
    return {"mad_lsb": round(float(np.abs(d).mean()), 4), "max_lsb": int(np.abs(d).max()),
            "psnr_db": round(10.0 * np.log10(255.0 ** 2 / max(mse, 1e-12)), 2),
            "against": "oracle (CPU fp32 restatement), same frame / weights / noise, 512x512 4-step ControlNet; "
                       "tolerance: mad <= 1.5 LSB, PSNR >= 38 dB"}


def api_leg(frames_host, n_frames=None, batch=5, lanes=2):
    """The drop-in class end to end: PIL in -> worker process -> PIL out through `VideoSDPipeline.remote(...)`
    (what diffusert/server.py:108 awaits), one frame at a time and as a stream the worker may coalesce."""
    import asyncio

    from PIL import Image

    from videosd_amd.pipeline import VideoSDPipeline

    opts = dict(prompt="pixar, cg", height=H, width=W, strength=STRENGTH, steps=LCM_STEPS, controlnet_scale=1.0, seed=23)
    imgs = [Image.fromarray(f, "RGB") for f in frames_host]
    w = VideoSDPipeline.remote(model="SimianLuo/LCM_Dreamshaper_v7", controlnet="lllyasviel/control_v11p_sd15_canny",
                               device=0, batch=batch, lanes=lanes, shm_slots=(lanes + 1) * batch + 4, call_timeout=600.0)
    n_frames = n_frames or 32 * batch
    try:
        # plans + graphs of every (batch size, lane) the stream will use: what a server does once at start-up
        w.method("warm_up")(batches=tuple(range(1, batch + 1)), lanes=lanes, **opts)
        lat = []
        for i in range(16):
            t0 = time.perf_counter()
            w.infer(imgs[i % len(imgs)], **opts)
            lat.append((time.perf_counter() - t0) * 1e3)

        async def stream(depth=int(os.environ.get("VSD_API_DEPTH", str((lanes + 1) * batch)))):
            sem = asyncio.Semaphore(depth)
            done = 0

            async def one(i):
                nonlocal done
                async with sem:
                    await w.infer.remote(imgs[i % len(imgs)], **opts)
                    done += 1

            t0 = time.perf_counter()
            await asyncio.gather(*[one(i) for i in range(n_frames)])
            return n_frames / (time.perf_counter() - t0)

        fps_stream = asyncio.run(stream())
        m = w.metrics()
        p50 = statistics.median(lat)
        return {"api_fps": round(fps_stream, 2), "api_p50_ms": round(p50, 2), "api_fps_one_at_a_time": round(1e3 / p50, 2),
                "api_stage_ms_p50": m.get("pipeline", {}).get("stage_ms_p50"),
                "api_frames_per_launch": m.get("frames_per_launch"),
                "api_note": "PIL 512x512 in -> VideoSDPipeline.remote worker process (shared-memory frame slots) -> PIL out; "
                            f"api_fps: {(lanes + 1) * batch} frames outstanding ({lanes} launches of up to {batch} on the GPU, one filling), the worker coalesces up to {batch} per launch"}
    finally:
        w.close()


def sessions_leg(frames_host, device_id):
    """Per-session state (server.py:90-93, 132-137, 163-197): what a prompt edit costs, and two sessions with different
    prompts AND frame sizes alternating frame by frame through the drop-in class (two lanes, one frame per launch)."""
    import numpy as np
    from PIL import Image

    from videosd_amd.pipeline import VideoSDPipeline

    p = VideoSDPipeline(model="SimianLuo/LCM_Dreamshaper_v7", controlnet="lllyasviel/control_v11p_sd15_canny", device=device_id)
    sa = dict(prompt="pixar, cg", height=H, width=W, strength=STRENGTH, steps=LCM_STEPS, controlnet_scale=1.0)
    sb = dict(prompt="an oil painting of a harbour at dusk", height=384, width=640, strength=0.5, steps=LCM_STEPS, controlnet_scale=1.2)
    imgs = [Image.fromarray(f, "RGB") for f in frames_host[:4]]
    for lane in (0, 1):  # both programs on both lanes: what a server's warm-up does
        for s_ in (sa, sb):
            p.collect_batch(p.submit_batch([imgs[0]], lane=lane, **s_))

    def stream(pattern, n):
        pend = []
        t0 = time.perf_counter()
        for i in range(n):
            if len(pend) == 2:
                p.collect_batch(pend.pop(0))
            pend.append(p.submit_batch([imgs[i % 4]], lane=i % 2, **pattern[i % len(pattern)]))
        while pend:
            p.collect_batch(pend.pop(0))
        return n / (time.perf_counter() - t0)

    stream([sa, sb], 8)
    n_prep, n_enc = len(p._host_ms["prepare"]), len(p._host_ms["prompt"])
    one = stream([sa], 48)
    two = stream([sa, sb], 48)          # A on lane 0, B on lane 1
    two_x = stream([sa, sa, sb, sb], 48)  # every lane sees both sessions: a prompt install (one device copy) per launch
    # a prompt edit: encode (stand-in embeddings here; CLIP on the GPU with real weights) + K / V projections of 23 layers +
    # the absorbed weights of 16 layers on the GPU, then the first frame with it
    t0 = time.perf_counter()
    p._cache_prompt("pixar, cg, edited", prompt="pixar, cg, edited")
    build_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    p.infer(imgs[0], **dict(sa, prompt="pixar, cg, edited"))
    first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    p.infer(imgs[0], **dict(sa, prompt="pixar, cg, edited"))
    steady_ms = (time.perf_counter() - t0) * 1e3
    return {"two_sessions_fps": round(two, 2), "two_sessions_fps_lanes_shared": round(two_x, 2), "one_session_fps_same_harness": round(one, 2),
            "two_sessions_note": "VideoSDPipeline in process, one frame per launch, two lanes; session A 512x512 'pixar, cg', session B 640x384 "
                                 "another prompt / strength / controlnet_scale, alternating frame by frame; prepares / prompt builds during the "
                                 f"timed streams: {len(p._host_ms['prepare']) - n_prep} / {len(p._host_ms['prompt']) - n_enc - 1}",
            "prompt_change_ms": round(build_ms + max(first_ms - steady_ms, 0.0), 2),
            "prompt_change_detail_ms": {"build_constants": round(build_ms, 2), "first_frame": round(first_ms, 2), "steady_frame": round(steady_ms, 2)}}


def run_rank(args):
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and "WORLD_SIZE" in os.environ and rank == 0:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}\n")
    if not args.dry_run and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.dry_run and os.environ.get("VSD_DRYRUN_FAIL_RANK") == str(rank):
        raise SystemExit(7)  # (tests/test_bench_launcher.py: a rank that dies before the rendezvous)
    dist = None
    backend = os.environ.get("VSD_DIST_BACKEND", "gloo" if args.dry_run else "nccl")
    if world > 1:
        import torch.distributed as dist

        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that never arrives must not hold the others for the default 10-30 minutes
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("VSD_PG_TIMEOUT", "300")))
        if os.environ.get("VSD_SHARE_GPU"):
            local = 0
        # RCCL over xGMI.  VSD_DIST_BACKEND=gloo + VSD_SHARE_GPU=1 exist only to walk this multi-rank code path on a
        # single-GPU box (both ranks on cuda:0, collectives on host copies); the driver never sets them.
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=pg_timeout)
        else:
            if not args.dry_run:
                torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)
    ranks_seen = 1
    if dist is not None:
        seen = [None] * world
        dist.all_gather_object(seen, (rank, local))
        ranks_seen = len({r for r, _ in seen})

    if args.dry_run:
        return dry_run(args, dist, rank, world, ranks_seen)

    eng, ops, weights = build_engine(local)
    # the four launch streams must sit on four different command-processor pipes: four frame-like kernel chains at once
    # against one alone (~1.0; >= 2 would mean two lanes take turns -- the "lottery" of rounds 1-3)
    pipes_check = ops.pool_check()
    tuning = os.path.join(ROOT, "profiles", "tuning_mi355x.json")
    if not args.retune:
        ops.load_tuning(tuning)  # per-shape (tile, split-K, pipeline) choices found by HipOps.tune_conv on an MI355X

    # prompt embeddings: produced on rank 0, broadcast over RCCL/xGMI
    text = torch.zeros(77, 768, dtype=torch.float16, device=ops.device)
    if rank == 0:
        text.copy_((torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half()[0])
    if dist is not None:
        from videosd_amd.dispatch import broadcast_prompt

        torch.cuda.current_stream().synchronize()
        hdr = {"epoch": 1, "height": H, "width": W, "steps": LCM_STEPS, "strength": STRENGTH, "controlnet_scale": 1.0, "seed": 23}
        if backend == "nccl":
            text, _ = broadcast_prompt(text if rank == 0 else None, hdr, src=0, device=ops.device)
        else:
            host, _ = broadcast_prompt(text.cpu() if rank == 0 else None, hdr, src=0, device=torch.device("cpu"))
            text.copy_(host)
        torch.cuda.current_stream().synchronize()
    eng.set_text_embeds(text)
    # throughput configuration: several launches in flight, one launch lane each.  Every engine is captured both ways (the
    # ControlNet encoder on the lane's side stream / everything on the lane's own stream) and the rule is the drop-in class's
    # (VideoSDPipeline._overlap_now): the side stream is used while at most two launches are in flight -- with three or four
    # the side streams ARE the other lanes' streams.  The single-frame latency below therefore runs with it.
    B = max(1, args.batch)
    eng.overlap_launch = args.slots < 3
    # kernel choices: with three or four lanes the forms that cost least when four lanes are busy, else the forms that are
    # fastest alone (the drop-in class's rule, VideoSDPipeline._engine_for); the latency leg below always takes the latter
    eng.tune_for_lanes = args.slots >= 3 and B > 1 and not os.environ.get("VSD_NO_LANE_TUNING")
    t_prep = time.perf_counter()
    plan = eng.prepare(H, W, LCM_STEPS, STRENGTH, controlnet_scale=1.0, use_controlnet=True, batch=B)
    launches_B = {"two_stream_form": eng.launches_by_kind()[0], "one_stream_form": eng.launches_by_kind(serial=True)[0]}
    prepare_ms = (time.perf_counter() - t_prep) * 1e3
    # a slider step (strength / controlnet_scale): device constants only, same captured graph
    t_upd = time.perf_counter()
    eng.update_options(0.62, 1.05)
    eng.update_options(STRENGTH, 1.0)
    update_options_ms = (time.perf_counter() - t_upd) * 1e3 / 2
    # frames are independent (the reference resets its RNG per frame): keep `slots` of them in flight per GPU,
    # each with its own buffers, streams and hipGraph, sharing the weight replica
    engines = [eng]
    for _ in range(max(1, args.slots) - 1):
        sl = eng.make_slot()
        sl.overlap_launch = eng.overlap_launch
        sl.tune_for_lanes = eng.tune_for_lanes
        sl.prepare(H, W, LCM_STEPS, STRENGTH, controlnet_scale=1.0, use_controlnet=True, batch=B)
        engines.append(sl)

    # this rank's shard of the synthetic stream, resident in HBM before the timed region
    nres = max(12, B)
    frames_host = synthetic_frames(nres * world, H, W)[rank::world]
    frames_dev = ops.to_device(torch.from_numpy(frames_host))

    def one_frame(i, pool=None):
        """launch i: the next B frames of this rank's shard (device-to-device, on the slot's own stream), one graph replay"""
        pool = pool or engines
        e = pool[i % len(pool)]
        nb = e.plan["batch"]
        k = (i * nb) % nres
        e.ops.copy_(e.frame_u8, (frames_dev[k:k + nb] if k + nb <= nres else frames_dev[:nb]).view_as(e.frame_u8))
        e.launch()

    def sync_all(pool=None):
        for e in (pool or engines):
            e.ops.synchronize()

    n_launch = -(-args.steps // B)  # K frames = ceil(K / B) launches (a ragged last launch still does B frames of work)
    for i in range(max(-(-args.warmup // B), len(engines))):  # W warm-up frames, and at least one replay of every slot's graph
        one_frame(i)
    sync_all()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_launch):
        one_frame(i)
    sync_all()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=ops.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    fps = world * args.steps / dt
    # the same plan timed for >= 2 s (the driver's --steps 20 is 0.17 s: 4 launches): a cross-check inside the line itself
    n_long = max(n_launch, int(2.2 / max(dt / n_launch, 1e-4)) + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_long):
        one_frame(i)
    sync_all()
    torch.cuda.synchronize()
    dt_long = time.perf_counter() - t0
    fps_long = n_long * B / dt_long  # (this rank's; reported for one GPU)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    extras = not args.no_extras and world == 1
    # ---- end to end: host u8 in -> host u8 out INSIDE the timed region (pinned staging, H2D / D2H on the launch's own
    #      stream), same frames per launch and launches in flight.  PCIe inclusive: reported beside `value`, never as it.
    fps_e2e = None
    got_batched = None
    if extras:
        # frame 0 of one launch of the TIMED plan (B frames): its image is compared with the oracle's below (`parity_batched`)
        got_batched = engines[0].infer_u8(np.ascontiguousarray(frames_host[:B]) if B > 1 else frames_host[0])
        got_batched = (got_batched[0] if B > 1 else got_batched).copy()
        nl = max(2 * len(engines), n_launch)
        fb = [np.ascontiguousarray(frames_host[(i * B) % (nres - B + 1):(i * B) % (nres - B + 1) + B]) for i in range(4)]
        for i in range(len(engines)):
            engines[i].submit_u8(fb[i % 4] if B > 1 else fb[i % 4][0])
        t1 = time.perf_counter()
        for i in range(nl):
            e = engines[i % len(engines)]
            e.collect_u8()
            e.submit_u8(fb[i % 4] if B > 1 else fb[i % 4][0])
        fps_e2e = nl * B / (time.perf_counter() - t1)
        for e in engines:
            e.collect_u8()

    # ---- p50 per-frame latency: host u8 in -> host u8 out (PCIe inclusive), (a) one frame in flight,
    #      (b) under the benchmark's load (`slots` frames in flight): submit -> that frame's u8 on the host
    def loaded_latency(pool, nb):
        """ms from submit (host u8) to that frame's u8 on the host with len(pool) launches of nb frames kept in flight"""
        out_ms, t_sub = [], {}
        nl = 6 * len(pool)
        for i in range(nl + len(pool)):
            e = pool[i % len(pool)]
            if i >= len(pool):  # the frame submitted len(pool) iterations ago on this slot
                e.ops.download(e.out_u8)
                out_ms.append((time.perf_counter() - t_sub[i - len(pool)]) * 1e3)
            if i < nl:
                t_sub[i] = time.perf_counter()
                e.ops.upload(e.frame_u8, torch.from_numpy(frames_host[:nb] if nb > 1 else frames_host[0]).view_as(e.frame_u8))
                e.launch()
        sync_all(pool)
        return out_ms

    lat_loaded = loaded_latency(engines, B) if len(engines) > 1 else []
    eng.overlap_launch = True
    eng.tune_for_lanes = False
    # (the launch sequence the drop-in class uses for a lone frame: ControlNet encoder on the lane's side stream, nothing else
    #  there -- `use_side_stream` measures level since the launch streams own their pipes: 47.8 vs 47.6 launches/s)
    plan1 = eng.prepare(H, W, LCM_STEPS, STRENGTH, controlnet_scale=1.0, use_controlnet=True, batch=1)
    launches_1 = {"two_stream_form": eng.launches_by_kind()[0], "one_stream_form": eng.launches_by_kind(serial=True)[0]}
    lat = []
    got0 = None
    for i in range(min(30, max(5, args.steps))):
        t1 = time.perf_counter()
        o = eng.infer_u8(frames_host[i % nres])
        lat.append((time.perf_counter() - t1) * 1e3)
        if i % nres == 0:
            got0 = o
    p50 = statistics.median(lat)

    # ---- other frames-per-launch operating points of the same engine: 1 (x3 in flight: round 1's first bench lines),
    #      3 (x2: round 1's final / round 2's earlier headline) and 8 (x2): throughput against frames in flight
    fps_b1 = None
    fps_by_b = {}
    lat_loaded_1x4 = []
    if extras:
        def throughput_at(b, nslots, overlap=False, loaded=None):
            pool = [eng] + list(engines[1:nslots])
            while len(pool) < nslots:
                pool.append(eng.make_slot())
            for e in pool:
                e.tune_for_lanes = nslots >= 3 and b > 1 and not os.environ.get("VSD_NO_LANE_TUNING")
                e.overlap_launch = overlap and nslots < 3
                e.prepare(H, W, LCM_STEPS, STRENGTH, controlnet_scale=1.0, use_controlnet=True, batch=b)
            for i in range(2 * nslots):
                one_frame(i, pool)
            sync_all(pool)
            t1 = time.perf_counter()
            nn = max(4 * nslots, args.steps // (2 * b))
            for i in range(nn):
                one_frame(i, pool)
            sync_all(pool)
            dt1 = time.perf_counter() - t1
            nn = max(nn, int(1.0 / max(dt1 / nn, 1e-4)) + 1)  # ... and again for >= 1 s (8 launches are 0.5 s of pipeline fill)
            t1 = time.perf_counter()
            for i in range(nn):
                one_frame(i, pool)
            sync_all(pool)
            got = nn * b / (time.perf_counter() - t1)
            if loaded is not None:
                loaded.extend(loaded_latency(pool, b))
            return got

        # one frame per launch (BASELINE configs[1] as worded): 3 lanes (rounds 1-3) and 4 -- the four launch streams are four
        # hardware queues on four command-processor pipes (ops.HipOps), so four independent frames run side by side
        fps_b1_3 = throughput_at(1, 3)
        fps_b1_4 = throughput_at(1, 4, loaded=lat_loaded_1x4)
        fps_b1 = max(fps_b1_3, fps_b1_4)
        fps_by_b = {"1x3": round(fps_b1_3, 2), "1x4": round(fps_b1_4, 2)}
        for b in (3, 8):
            if b != B:
                fps_by_b[f"{b}x2"] = round(throughput_at(b, 2, True), 2)
        fps_by_b[f"{B}x{len(engines)}"] = round(fps, 2)

    # ---- the same graph without the ControlNet tower (engine extension)
    for e in engines:
        e.overlap_launch = args.slots < 3
        e.tune_for_lanes = args.slots >= 3 and B > 1 and not os.environ.get("VSD_NO_LANE_TUNING")
        e.prepare(H, W, LCM_STEPS, STRENGTH, use_controlnet=False, batch=B)
    for i in range(3):
        one_frame(i)
    sync_all()
    t1 = time.perf_counter()
    nn = max(10, args.steps // 3 // B)
    for i in range(nn):
        one_frame(i)
    sync_all()
    fps_nocn = nn * B / (time.perf_counter() - t1)

    if args.save_tuning:
        ops.save_tuning(tuning)

    # ---- dominant kernel (implicit-GEMM conv) against the MFMA roofline: HIP events around every launch of
    #      one eager pass of the same program on the same stream
    eng.overlap_controlnet = False  # (one stream: a launch's event bracket must not include another stream's kernel beside it)
    eng.tune_for_lanes = False      # (... and each layer in the form that is fastest ALONE: the bracket times a kernel alone)
    eng.prepare(H, W, LCM_STEPS, STRENGTH, controlnet_scale=1.0, use_controlnet=True, use_graph=False, batch=B)
    eng.overlap_controlnet = True
    one_frame(0, [eng])
    ops.synchronize()
    ops.profile_begin()
    one_frame(1, [eng])
    ops.synchronize()
    st = ops.profile_end()
    cg = st["conv_gemm"]
    ovh_ms = ops.profile_overhead_ms()  # an empty HIP-event bracket: the timing's own cost per launch
    cg_ms = max(cg["ms"] - cg["launches"] * ovh_ms, 1e-6)
    achieved = cg["flops"] / (cg_ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_conv_gemm.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # ---- what the families OCCUPY of the chip while the TIMED program runs (5 frames per launch x 4 lanes, throughput-mode kernel
    #      forms): the sum of workgroup lives per family under captured-graph replay on all lanes, from the instrumented build
    #      (scripts/wg_cu_time.py; a constant of the committed profile, not measured in this run -- like `traffic`)
    in_situ = None
    cut_name = next((n for n in ("round6_wg_cu_time_5x4.txt", "round5f_wg_cu_time_5x4.txt") if os.path.exists(os.path.join(ROOT, "profiles", n))), "")
    cut = os.path.join(ROOT, "profiles", cut_name)
    if cut_name:
        try:
            runs = json.loads(open(cut).read().strip().splitlines()[-1])["runs"]
            timed = next(r for r in runs if r["mode"] == 1 and r["lanes"] == 4)
            alone = next(r for r in runs if r["mode"] == 0 and r["lanes"] == 1)
            per_cu_peak = MFMA_PEAK_TFLOPS / 256.0
            fams = [f for f, v in timed["families"].items() if v["wg_ms_per_frame"] > 0]
            in_situ = {"source": f"profiles/{cut_name} (scripts/wg_cu_time.py with the -DVSD_WG_TIMELINE build: every workgroup adds its life to a per-family "
                                 "counter; a constant of the committed profile)",
                       "program": f"{timed['frames_per_launch']} frames per launch x {timed['lanes']} lanes, throughput-mode kernel forms, captured graphs",
                       "wg_ms_per_frame": {f: timed["families"][f]["wg_ms_per_frame"] for f in fams},
                       "wg_ms_per_frame_fastest_alone_forms_one_lane": {f: alone["families"][f]["wg_ms_per_frame"] for f in fams},
                       "tflop_per_wg_second": {f: timed["families"][f]["tflop_per_wg_second"] for f in fams if timed["families"][f]["tflop_per_wg_second"]},
                       "frac_of_per_cu_mfma_peak_while_resident": {f: round(timed["families"][f]["tflop_per_wg_second"] / per_cu_peak, 4) for f in fams
                                                                   if timed["families"][f]["tflop_per_wg_second"]},
                       "per_cu_peak_tflops": round(per_cu_peak, 3),
                       "workgroups_resident_on_average": round(timed["wg_ms_per_frame_total"] / timed["wall_ms_per_frame"], 1),
                       "mean_resident_waves_per_simd": timed["mean_resident_waves_per_simd"],
                       "fps_of_the_instrumented_run": timed["fps"]}
        except Exception as e:  # reporting only
            in_situ = {"error": f"{type(e).__name__}: {e}"}
    # ---- the same fraction from the committed rocprofv3 kernel trace of one eager single-stream pass, every layer a launch of its
    #      own (scripts/layer_table.sh): sum of the conv family's FLOP / sum of its kernel nanoseconds, reducers included -- a
    #      constant of the committed profile, reproducible from profiles/ alone
    frac_trace = trace_src = None
    for name in ("round6_layer_table_batch%d.txt" % B, "round5f_layer_table_batch%d.txt" % B):
        lt = os.path.join(ROOT, "profiles", name)
        if os.path.exists(lt):
            try:
                import ast

                conv_ms = float(ast.literal_eval(open(lt).read().splitlines()[1])["conv"])
                frac_trace = round(cg["flops"] / (conv_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
                trace_src = f"profiles/{name}: conv family {conv_ms} ms of kernels per pass (rocprofv3 --kernel-trace, reducers included)"
                break
            except Exception:
                pass
    roofline = {"bound": "mfma", "kernel": "conv_gemm_kernel (implicit-GEMM conv/linear, all shapes of one frame)",
                "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "frac_trace": frac_trace, "frac_trace_source": trace_src, "traffic": traffic,
                "traffic_source": "profiles/pmc_conv_gemm.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/collect_profiles.sh; "
                                  "a constant of the committed profile, not measured in this run)",
                "launches_per_pass": cg["launches"], "frames_per_pass": B, "avg_launch_us": round(cg_ms * 1e3 / max(cg["launches"], 1), 2),
                "avg_launch_us_raw_events": round(cg["ms"] * 1e3 / max(cg["launches"], 1), 2),
                "event_bracket_overhead_us": round(ovh_ms * 1e3, 2),
                "flop_per_launch_avg": cg["flops"] / max(cg["launches"], 1),
                "families_ms_per_pass": {k: round(v["ms"], 3) for k, v in st.items()},
                "in_situ": in_situ}

    out = {
        "metric": METRIC,
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "value_long": round(fps_long, 3), "value_long_seconds": round(dt_long, 2),
        "vs_baseline": None, "dtype": "f16 (fp32 accumulate)", "data": "synthetic", "ranks_seen": ranks_seen,
        "config": {"workload": f"{B} frames per launch x {len(engines)} launch lanes in flight (a stream's independent frames coalesced along GEMM M; "
                               "the batch=1-per-launch figures of the same run are fps_one_frame_per_launch and p50_latency_ms) of "
                               "SD1.5 512x512 LCM 4-step img2img, every frame denoised independently, ControlNet-canny + TAESD "
                               "(BASELINE configs[1], reference-faithful: the reference always runs ControlNet)",
                   "frames_per_rank": args.steps, "sharding": f"round-robin frames over {world} GPU(s)",
                   "frames_per_launch": B, "launches_in_flight_per_gpu": len(engines),
                   "timesteps": plan["timesteps"],
                   # kernels one replay issues (Engine.launches_by_kind: reducers and second GroupNorm kernels counted, a pair / group of
                   # calls sharing a grid once): the timed program runs the one-stream form (lanes >= 3), a lone frame the two-stream form
                   "kernel_launches_per_graph_replay": launches_B["one_stream_form" if len(engines) >= 3 else "two_stream_form"],
                   # (p50_latency_ms is a lone frame in the two-stream form; 1 x 4 runs the one-stream form: kernel_launches_by_form)
                   "kernel_launches_per_single_frame_graph": launches_1["two_stream_form"],
                   "kernel_launches_by_form": {"frames_per_launch_%d" % B: launches_B, "one_frame": launches_1},
                   "recorded_ops": {"frames_per_launch_%d" % B: plan["n_ops"], "one_frame": plan1["n_ops"]}},
        "p50_latency_ms": round(p50, 3),
        "p50_latency_ms_under_load": round(statistics.median(lat_loaded), 3) if lat_loaded else None,
        "p50_latency_ms_under_load_one_frame_per_launch_x4": round(statistics.median(lat_loaded_1x4), 3) if lat_loaded_1x4 else None,
        "fps_one_frame_per_launch": round(fps_b1, 3) if fps_b1 else None,
        "fps_by_frames_per_launch_x_launches_in_flight": fps_by_b or None,
        "fps_end_to_end": round(fps_e2e, 3) if fps_e2e else None,
        "fps_without_controlnet": round(fps_nocn, 3),
        "launch_streams": {"lanes": "lane l = launch stream l mod 4 (CU-masked: own hardware queue), side branch on (l + 2) mod 4",
                           "four_chains_vs_one": round(pipes_check, 3),
                           "graphs_per_launch": plan.get("graphs"), "event_edges_per_launch": plan.get("edges")},
        "prepare_ms": round(prepare_ms, 1), "update_options_ms": round(update_options_ms, 2),
        "frame_roofline": {"algorithmic_tflop_per_frame": 4.623, "mfma_frac": round(4.623 * fps / world / MFMA_PEAK_TFLOPS, 4)},
        "roofline": roofline,
    }
    if extras:
        try:
            out.update(sessions_leg(frames_host, local))
        except Exception as e:  # reporting only; never lose the measured line
            out["two_sessions_fps"] = None
            out["two_sessions_note"] = f"failed: {type(e).__name__}: {e}"
    if extras and not args.no_api:
        # drop the bench's own engines first: the API worker is a second process with its own weight replica
        try:
            out.update(api_leg(frames_host, batch=B, lanes=int(os.environ.get("VSD_API_LANES", str(len(engines))))))
        except Exception as e:  # reporting only; never lose the measured line
            out["api_fps"] = None
            out["api_note"] = f"failed: {type(e).__name__}: {e}"
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"], ref0 = cpu_baseline(weights, text.cpu(), frames_host[0])
            if got0 is not None:
                out["parity"] = image_parity(got0, ref0)  # the one-frame plan of the latency leg
            if got_batched is not None:
                out["parity_batched"] = dict(image_parity(got_batched, ref0), frames_per_launch=B)  # the timed plan
        except Exception as e:  # the baseline is reporting only; never lose the measured line
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"failed: {e}"}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def dry_run(args, dist, rank, world, ranks_seen):
    """The multi-rank protocol of `run_rank` with the GPU taken out: rank 0's prompt is broadcast, every rank 'does' K
    frames, barrier + max-over-ranks timing, rank 0 prints the line.  CPU test of the plumbing only."""
    import torch

    from videosd_amd.dispatch import broadcast_prompt

    text = (torch.randn(77, 768, generator=torch.Generator().manual_seed(7)) * 0.5).half() if rank == 0 else None
    checksum = None
    if dist is not None:
        buf, hdr = broadcast_prompt(text, {"epoch": 1, "steps": LCM_STEPS}, src=0, device=torch.device("cpu"))
        sums = [None] * world
        dist.all_gather_object(sums, float(buf.float().sum()))
        checksum = sums
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "dry_run": True, "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "ranks_seen": ranks_seen, "prompt_checksums": checksum, "scaling": "weak"}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    main()
