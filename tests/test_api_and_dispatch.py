"""Drop-in surface (reference videopipeline.py:11-128, server.py:104-143,317-321) and the multi-GPU dispatch
logic, on CPU: signatures/defaults, crop+resize, schedule helpers, awaitable `.infer.remote`, round-robin
sharding with in-order release and drop-if-busy, RCCL-broadcast logic with gloo standing in (world_size 2)."""
import asyncio
import time
import inspect
import os
import socket
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from videosd_amd import lcm
from videosd_amd.dispatch import FrameDispatcher, RemotePipeline, owner_of, shard_indices
from videosd_amd.pipeline import VideoPipeline, VideoSDPipeline, center_crop_resize

FAKE = "helpers_fake_pipeline:FakePipeline"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["PYTHONPATH"] = os.path.dirname(os.path.abspath(__file__)) + os.pathsep + os.environ.get("PYTHONPATH", "")


def test_infer_signature_matches_reference():
    sig = inspect.signature(VideoSDPipeline.infer)
    got = [(n, p.default) for n, p in sig.parameters.items()][2:]
    assert got == [("prompt", ["pixar, cg"]), ("height", 360), ("width", 640), ("strength", 0.4), ("steps", 20),
                   ("guidance_scale", 7.5), ("ref", False), ("style_fidelity", 0.0), ("controlnet", False), ("seed", 42),
                   ("controlnet_scale", 1)]
    assert VideoPipeline is VideoSDPipeline
    assert {"load_model", "compile_model", "infer", "remote"} <= set(dir(VideoSDPipeline))


def test_constructor_requires_model_and_controlnet_and_a_gpu():
    with pytest.raises(KeyError):
        VideoSDPipeline(gpus=4, compile=False)
    if not torch.cuda.is_available():  # no CPU fallback: the product path fails loudly without the GPU
        with pytest.raises(RuntimeError, match="GPU|libvsd"):
            VideoSDPipeline(model="m", controlnet="c")


def test_center_crop_resize_matches_oracle_restatement():
    from oracle.pipeline import center_crop_resize as ref

    rng = np.random.default_rng(0)
    for (w, h), (tw, th) in [((640, 480), (512, 512)), ((300, 500), (640, 360)), ((768, 432), (768, 432))]:
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
        assert np.array_equal(np.asarray(center_crop_resize(img, tw, th)), np.asarray(ref(img, tw, th)))


def test_product_schedule_matches_golden_and_oracle():
    import json

    from oracle.scheduler import LCMSchedulerOracle, w_embedding

    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lcm_scheduler.json")))
    for c in g["timesteps"]:
        assert lcm.lcm_timesteps(c["strength"], c["steps"]) == c["timesteps"]
    ref = np.frombuffer(bytes.fromhex(g["w_embedding_7p5_f32_hex"]), dtype=np.float32).reshape(1, 256)
    assert np.array_equal(lcm.w_embedding(7.5, 256).numpy(), ref)
    assert np.array_equal(lcm.alphas_cumprod().numpy(),
                          np.frombuffer(bytes.fromhex(g["alphas_cumprod_f32_hex"]), dtype=np.float32))
    sch = LCMSchedulerOracle()
    sch.set_timesteps(0.6, 4)
    plan = lcm.LCMSchedule(0.6, 4)
    for i, t in enumerate(plan.timesteps):
        sa, sb, cs, co, sap, sbp = plan.step_coef(i)
        a = sch.alphas_cumprod[t]
        assert sa == float(a.sqrt()) and sb == float((1 - a).sqrt())
        rs, ro = sch.scalings(torch.tensor(t))
        assert cs == float(rs) and co == float(ro)
    with pytest.raises(ValueError):
        lcm.lcm_timesteps(0.01, 4)


def test_host_noise_follows_the_reference_rng_contract():
    import json

    from videosd_amd.engine import Engine

    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lcm_scheduler.json")))
    torch.manual_seed(1234)
    before = torch.get_rng_state()
    nz, _ = Engine.host_noise(4, 64, 64)
    assert torch.equal(before, torch.get_rng_state())  # the caller's RNG stream is left untouched
    assert nz.shape == (5, 4, 64 * 64)
    assert float(nz[0].sum()) == pytest.approx(g["rng"]["23"]["draw0_sum"], rel=1e-6)
    assert nz[0, 0, :4].tolist() == g["rng"]["23"]["draw0_first4"]
    assert float(nz[1].sum()) == pytest.approx(g["rng"]["23"]["draw1_sum"], rel=1e-6)
    assert Engine.host_noise(1, 8, 8)[0].shape == (2, 4, 64)
    # reference-only mode: every step draws the reference latents' noise first (lcm_reference_pipeline.py:861-863)
    d, rd = Engine.host_noise(2, 8, 8, ref=True)
    torch.manual_seed(0)
    torch.default_generator.set_state(torch.Generator(device="cpu").get_state())
    seq = [torch.randn(1, 4, 8, 8) for _ in range(5)]
    assert torch.equal(d, torch.cat([seq[0], seq[2], seq[4]]).reshape(3, 4, 64)) and torch.equal(rd, torch.cat([seq[1], seq[3]]).reshape(2, 4, 64))


def test_sharding_is_strict_round_robin():
    assert shard_indices(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((shard_indices(37, r, 8) for r in range(8)), [])) == list(range(37))
    assert [owner_of(k, 3) for k in range(6)] == [0, 1, 2, 0, 1, 2]


def _img(v, size=(16, 12)):
    return Image.fromarray(np.full((size[1], size[0], 3), v, dtype=np.uint8), "RGB")


def test_remote_pipeline_is_awaitable_like_a_ray_actor():
    p = RemotePipeline(factory=FAKE, model="m", controlnet="c", device=3)
    try:
        async def go():
            a = p.infer.remote(_img(10), height=12, width=16)
            b = p.infer.remote(_img(20), height=12, width=16)
            return await a, await b

        a, b = asyncio.run(go())
        assert np.asarray(a)[1, 1, 0] == 245 and np.asarray(b)[1, 1, 0] == 235 and np.asarray(a)[0, 0, 0] == 3
        assert np.asarray(p.infer(_img(30), height=12, width=16))[1, 1, 0] == 225  # synchronous form

        async def bad():
            await p.infer.remote(_img(1), strength=-1.0)

        with pytest.raises(RuntimeError, match="negative strength"):
            asyncio.run(bad())
    finally:
        p.close()
    with pytest.raises(KeyError):
        RemotePipeline(factory=FAKE, gpus=2)


def test_dispatcher_round_robin_in_order_release_and_drop_if_busy():
    ps = [RemotePipeline(factory=FAKE, model="m", controlnet="c", device=i, delay=0.15 if i == 0 else 0.01) for i in range(2)]
    try:
        async def go():
            d = FrameDispatcher(ps, mode="in_order")
            t = [d.submit(_img(10 * (k + 1)), height=12, width=16) for k in range(4)]
            # frames 0,1 go to workers 0,1; frames 2,3 arrive while both are busy -> dropped
            assert t == [0, 1, None, None] and d.dropped == 2
            out = [await d.next_result() for _ in range(2)]
            # worker 1 finishes first, but release is in submission order
            assert [o[0] for o in out] == [0, 1]
            assert [int(np.asarray(o[1])[0, 0, 0]) for o in out] == [0, 1]
            assert [int(np.asarray(o[1])[1, 1, 0]) for o in out] == [245, 235]
            # latest-wins mode shows the newest finished frame (server.py:117)
            d2 = FrameDispatcher(ps, mode="latest")
            d2.submit(_img(1), height=12, width=16)
            d2.submit(_img(2), height=12, width=16)
            await asyncio.sleep(0.4)
            tk, img = await d2.next_result()
            assert tk == 1 and d2.pending == 0
            return True

        assert asyncio.run(go())
    finally:
        for p in ps:
            p.close()


def test_worker_coalesces_queued_frames_into_batched_launches():
    """RemotePipeline(batch=3): frames that are already waiting behind the one being served ride in one `infer_batch`
    launch (same options only); a lone frame is served at once; every caller still gets its own frame back."""
    p = RemotePipeline(factory=FAKE, model="m", controlnet="c", device=1, delay=0.25, batch=3)
    try:
        async def go():
            futs = [p.infer.remote(_img(10 * (k + 1)), height=12, width=16) for k in range(5)]
            other = p.infer.remote(_img(90), height=12, width=16, strength=0.7)  # different options: its own launch
            return [await f for f in futs], await other

        outs, other = asyncio.run(go())
        vals = [int(np.asarray(o)[1, 1, 0]) for o in outs]
        assert vals == [245, 235, 225, 215, 205]  # everyone got their own frame
        # tag = size of the batch the frame rode in; a frame served alone goes through the plain `infer` (untagged pixel)
        sizes = [t if t <= 3 else 1 for t in (int(np.asarray(o)[0, 0, 1]) for o in outs)]
        # frame 0 was taken alone or with whatever had already arrived; later ones were coalesced (never more than 3)
        assert max(sizes) <= 3 and sum(1 for s_ in sizes if s_ > 1) >= 2, sizes
        assert int(np.asarray(other)[1, 1, 0]) == 165 and int(np.asarray(other)[0, 0, 1]) in (1, 165)  # a launch of its own
        # two launches in flight: consecutive launches alternate between the two engine lanes
        assert {int(np.asarray(o)[0, 0, 2]) for o in outs} == {0, 1}
        # an error inside a batched launch reaches every caller of that launch
        async def bad():
            a = p.infer.remote(_img(1), strength=-1.0)
            b = p.infer.remote(_img(2), strength=-1.0)
            res = []
            for f in (a, b):
                try:
                    await f
                    res.append(None)
                except RuntimeError as e:
                    res.append(str(e))
            return res

        errs = asyncio.run(bad())
        assert all(e and "negative strength" in e for e in errs), errs
    finally:
        p.close()


def test_worker_with_three_lanes_keeps_three_launches_in_flight_and_answers_in_order():
    """RemotePipeline(lanes=3): consecutive launches rotate over three engine lanes, at most three are on the 'GPU' at once,
    and every caller still gets its own frame, in request order."""
    p = RemotePipeline(factory=FAKE, model="m", controlnet="c", device=1, delay=0.15, batch=2, lanes=3)
    try:
        async def go():
            futs = [p.infer.remote(_img(10 * (k + 1)), height=12, width=16) for k in range(12)]
            return [await f for f in futs]

        t0 = time.time()
        outs = asyncio.run(go())
        dt = time.time() - t0
        assert [int(np.asarray(o)[1, 1, 0]) for o in outs] == [255 - 10 * (k + 1) for k in range(12)]
        assert {int(np.asarray(o)[0, 0, 2]) for o in outs} == {0, 1, 2}  # all three lanes were used
        assert dt < 12 * 0.15  # launches overlapped (one at a time would take >= 6 launches x 0.15 s + the first frame alone)
    finally:
        p.close()


@pytest.mark.timeout(120)
def test_large_frames_queued_ahead_do_not_deadlock_the_pipe():
    """Real frames (~0.8 MB pickled) are far larger than the pipe buffers: several requests queued while the worker is
    busy and a result of the same size coming back must not block each other (requests leave through a writer thread)."""
    p = RemotePipeline(factory=FAKE, model="m", controlnet="c", device=0, delay=0.05, batch=3)
    try:
        async def go():
            big = [_img(10 + k, size=(512, 512)) for k in range(8)]
            futs = [p.infer.remote(im, height=512, width=512) for im in big]
            return [await asyncio.wait_for(f, timeout=60) for f in futs]

        outs = asyncio.run(go())
        assert [int(np.asarray(o)[5, 5, 0]) for o in outs] == [245 - k for k in range(8)]
    finally:
        p.close()


def test_dispatcher_depth_keeps_several_frames_per_worker():
    ps = [RemotePipeline(factory=FAKE, model="m", controlnet="c", device=i, delay=0.1, batch=2) for i in range(2)]
    try:
        async def go():
            d = FrameDispatcher(ps, mode="in_order", depth=2)
            t = [d.submit(_img(10 * (k + 1)), height=12, width=16) for k in range(5)]
            assert t == [0, 1, 2, 3, None] and d.dropped == 1  # two frames per worker, the fifth finds worker 0 full
            out = [await d.next_result() for _ in range(4)]
            assert [o[0] for o in out] == [0, 1, 2, 3]
            assert [int(np.asarray(o[1])[0, 0, 0]) for o in out] == [0, 1, 0, 1]  # frame k -> worker k mod 2
            assert [int(np.asarray(o[1])[1, 1, 0]) for o in out] == [245, 235, 225, 215]
            assert d.busy == [0, 0]
            return True

        assert asyncio.run(go())
    finally:
        for p in ps:
            p.close()


def test_frame_loop_example_with_a_stand_in_worker():
    """examples/frame_loop.py (the reference's recv/diffuse loop with a synthetic camera) end to end on a fake worker."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import frame_loop

    out = frame_loop.main(["--factory", FAKE, "--gpus", "2", "--fps", "40", "--seconds", "1.5", "--batch", "2", "--mode", "in_order",
                           "--height", "12", "--width", "16", "--worker-config", '{"delay": 0.02}'])
    assert out["offered"] >= 40 and out["processed"] + out["dropped"] == out["offered"]
    assert out["shown"] == out["processed"] and out["output_fps"] > 20 and out["p50_latency_ms"] < 500


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, q):
    import torch.distributed as dist

    from videosd_amd.dispatch import broadcast_prompt, shard_indices

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    emb = None
    hdr = None
    if rank == 0:
        emb = (torch.arange(77 * 768, dtype=torch.float32).reshape(77, 768) % 97 / 97).half()
        hdr = {"epoch": 3, "height": 512, "width": 512, "steps": 4, "strength": 0.6, "controlnet_scale": 2.0, "seed": 23}
    buf, h = broadcast_prompt(emb, hdr, src=0, device=torch.device("cpu"))
    mine = shard_indices(11, rank, world)
    # every rank "processes" its shard; results are gathered to check coverage and ordering
    out = [None] * world
    dist.all_gather_object(out, [(k, rank) for k in mine])
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing
    q.put((rank, float(buf.float().sum()), h, out, float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_sharding():
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    ref = float(((torch.arange(77 * 768, dtype=torch.float32).reshape(77, 768) % 97 / 97).half()).float().sum())
    for rank, s, h, out, tmax in res:
        assert s == ref and h["steps"] == 4.0 and h["controlnet_scale"] == 2.0 and h["epoch"] == 3.0
        flat = sorted(sum(out, []))
        assert [k for k, _ in flat] == list(range(11)) and all(r == k % 2 for k, r in flat)
        assert tmax == pytest.approx(0.2)


def test_plan_cache_respects_its_memory_budget():
    """VideoSDPipeline._trim_memory: over the budget, idle programs go least recently used first, then the idle slots of the
    program being extended; the program's root and anything with a launch in flight stay."""
    from collections import OrderedDict

    from videosd_amd.pipeline import VideoSDPipeline

    class E:
        def __init__(self, gb):
            self.gb, self.destroyed = gb, False

        def _destroy_graphs(self):
            self.destroyed = True

    p = VideoSDPipeline.__new__(VideoSDPipeline)
    p._plans, p._outstanding, p.evictions = OrderedDict(), [], 0

    def plan(*gbs):
        es = [E(g) for g in gbs]
        return {"root": es[0], "opts": (0.6, 1.0), "engines": {(i + 1, 0): e for i, e in enumerate(es)}}

    p._plans["old"], p._plans["busy"], p._plans["mid"], p._plans["cur"] = plan(6, 6), plan(5), plan(4), plan(3, 2, 2)
    p._outstanding = [p._plans["busy"]["root"], p._plans["cur"]["engines"][(2, 0)]]
    p._memory_used_and_limit = lambda before=None: (sum(e.gb for pl in p._plans.values() for e in pl["engines"].values()), 20)
    assert p._trim_memory(keep="cur") == 2 and list(p._plans) == ["busy", "mid", "cur"]   # 28 -> 16: "old" alone is enough
    p._memory_used_and_limit = lambda before=None: (sum(e.gb for pl in p._plans.values() for e in pl["engines"].values()), 9)
    cur = p._plans["cur"]
    idle_slot = cur["engines"][(3, 0)]
    assert p._trim_memory(keep="cur") == 2                                                # "mid", then cur's one idle slot
    assert list(p._plans) == ["busy", "cur"] and sorted(cur["engines"]) == [(1, 0), (2, 0)] and idle_slot.destroyed
    assert not cur["root"].destroyed and not p._plans["busy"]["root"].destroyed and p.evictions == 4
    assert p._trim_memory(keep="cur") == 0                                                # nothing idle left: allocation goes ahead


def test_side_stream_policy_is_a_function_of_the_launches_in_flight():
    """VideoSDPipeline._overlap_now: a launch runs its ControlNet encoder on the lane's side stream only while it has a
    command-processor pipe for it -- at most two launches in flight, all on lanes 0 / 1 (lane l's side stream is lane l + 2's
    own stream: videosd_amd/ops.py).  bench.py's engine legs state the same rule (`overlap_launch = slots < 3`)."""
    from videosd_amd.pipeline import VideoSDPipeline

    p = VideoSDPipeline.__new__(VideoSDPipeline)
    os.environ.pop("VSD_OVERLAP_CN", None)
    p._lanes_busy = []
    assert p._overlap_now(0) and p._overlap_now(1) and not p._overlap_now(2) and not p._overlap_now(3)
    p._lanes_busy = [0]
    assert p._overlap_now(1) and not p._overlap_now(2)
    p._lanes_busy = [0, 1]
    assert not p._overlap_now(0) and not p._overlap_now(2)   # a third launch: three lanes busy
    p._lanes_busy = [2]
    assert not p._overlap_now(0)                             # lane 0's side stream IS lane 2's stream


def test_remote_without_device_takes_the_next_gpu_like_a_ray_actor():
    """server.py:320-321 writes `VideoSDPipeline.remote(**config)` with no device: Ray's num_gpus=1 (videopipeline.py:11) gives
    every actor its own GPU.  Here the k-th worker created without `device` gets GPU k (mod the GPU count, resolved in the
    worker); a respawn keeps the dead worker's GPU; an explicit `device` does not consume an ordinal."""
    from videosd_amd import dispatch as D

    with RemotePipeline._auto_lock:
        RemotePipeline._auto_next = 0
    ws = [RemotePipeline(factory=FAKE, model="m", controlnet="c", wait=False) for _ in range(4)]
    fixed = RemotePipeline(factory=FAKE, model="m", controlnet="c", device=7, wait=False)
    late = RemotePipeline(factory=FAKE, model="m", controlnet="c", wait=False)
    try:
        tags = [int(np.asarray(w.infer(_img(10), height=12, width=16))[0, 0, 0]) for w in ws + [fixed, late]]
        assert tags == [0, 1, 2, 3, 7, 4]
        assert [w.device for w in ws] == ["auto:0", "auto:1", "auto:2", "auto:3"]
        ws[2] = ws[2].respawn()
        assert int(np.asarray(ws[2].infer(_img(10), height=12, width=16))[0, 0, 0]) == 2
        assert asyncio.run(_metrics_of(ws[1]))["device"] == 1
    finally:
        for w in ws + [fixed, late]:
            w.close()
    # the worker-side rule: ordinal modulo the GPUs the worker sees (this box: none -> a stand-in keeps its ordinal)
    n = torch.cuda.device_count()
    assert D.resolve_auto_device({"device": "auto:9"})["device"] == (9 % n if n else 9)
    assert D.resolve_auto_device({"device": 3})["device"] == 3 and "device" not in D.resolve_auto_device({})


async def _metrics_of(w):
    return await w.metrics.remote()


def test_a_prompt_list_is_one_prompt_or_refused():
    """lcm_controlnet.py:433-438 makes a list a BATCH of prompts; the per-frame path returns one image (videopipeline.py:126-128):
    a string and the one-element default are the same prompt, anything longer is refused by name (it used to be joined)."""
    one = VideoSDPipeline._one_prompt
    assert one("pixar, cg") == one(["pixar, cg"]) == one(("pixar, cg",)) == "pixar, cg"
    for bad in (["a", "b"], [], [3]):
        with pytest.raises(ValueError, match="lcm_controlnet.py:433-438"):
            one(bad)
