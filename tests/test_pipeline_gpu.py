"""End-to-end parity on the MI355X: the HIP engine (through the C-ABI) against the CPU oracle on the same
seeded weights, frame, text embeddings and noise draws.

Tolerances (fp16 storage + fp32 accumulation vs an fp32 oracle):
  TAESD-encoded latents rel-L2 <= 5e-3; final denoised latent rel-L2 <= 2e-2;
  output image mean |diff| <= 1.5 LSB and PSNR >= 38 dB."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _frame(h, w, seed=1):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = ((xx * 5 + yy * 3) % 256).astype(np.uint8)[..., None]
    return (base // 2 + grad // 2).astype(np.uint8)


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def _build(unet_cfg, cn_cfg, dev="cuda"):
    from videosd_amd import config as C
    from videosd_amd import weights as W

    wu = W.synthesize(W.unet_spec(unet_cfg), "unet.", device=dev)
    wc = W.synthesize(W.controlnet_spec(cn_cfg), "cn.", device=dev)
    wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device=dev)
    text = (torch.randn(77, unet_cfg.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5).half()
    return wu, wc, wv, text


def _cpu(w):
    return {k: v.cpu() for k, v in w.items()}


def _compare(eng, orc, frame, text, H, W, steps, cn, cn_scale=1.5):
    got = eng.infer_u8(frame)
    ref = np.asarray(orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=H, width=W, strength=0.6,
                               steps=steps, seed=23, controlnet_scale=cn_scale, use_controlnet=cn, keep_trace=True))
    h0, w0 = H // 8, W // 8
    x0 = eng.buffers["x0"][:, :4].float().cpu().reshape(h0, w0, 4).permute(2, 0, 1)
    ref_x0 = orc.trace["init_latents"][0]
    r0 = float((x0 - ref_x0).norm() / ref_x0.norm())
    den = eng.buffers["denoised"][:, :4].float().cpu().reshape(h0, w0, 4).permute(2, 0, 1)
    ref_den = orc.trace["denoised"][-1][0]
    r1 = float((den - ref_den).norm() / ref_den.norm())
    diff = np.abs(got.astype(int) - ref.astype(int))
    return r0, r1, float(diff.mean()), _psnr(got, ref), got


from golden_guard import CASES as GOLDEN_CASES, GOLDEN_FULLSIZE  # noqa: E402  (tests/test_oracle_golden.py guards the fixture's sources)


def _compare_golden(eng, frame, H, W, case):
    """The same numbers as `_compare`, against the oracle output stored by scripts/make_fullsize_golden.py (the two
    slowest full-size cases: 3 minutes of CPU oracle each; VSD_LIVE_ORACLE=1 runs the oracle instead): latents in fp16,
    every second row / column of the image."""
    got = eng.infer_u8(frame)
    with np.load(GOLDEN_FULLSIZE) as z:
        ref_half, ref_den = z[case + "_image_half"], torch.from_numpy(z[case + "_denoised"]).float()
        ref_x0 = torch.from_numpy(z[case + "_init_latents"]).float() if case + "_init_latents" in z.files else None
    h0, w0 = H // 8, W // 8
    r0 = 0.0
    if ref_x0 is not None:
        x0 = eng.buffers["x0"][:, :4].float().cpu().reshape(h0, w0, 4).permute(2, 0, 1)
        r0 = float((x0 - ref_x0).norm() / ref_x0.norm())
    den = eng.buffers["denoised"][:, :4].float().cpu().reshape(h0, w0, 4).permute(2, 0, 1)
    r1 = float((den - ref_den).norm() / ref_den.norm())
    half = got[::2, ::2]
    diff = np.abs(half.astype(int) - ref_half.astype(int))
    return r0, r1, float(diff.mean()), _psnr(half, ref_half), got


@pytest.fixture(scope="module")
def mini_setup():
    from oracle.pipeline import OraclePipeline
    from videosd_amd import config as C
    from videosd_amd.engine import Engine
    from videosd_amd.ops import HipOps

    wu, wc, wv, text = _build(C.MINI_UNET, C.MINI_CONTROLNET)
    eng = Engine(HipOps(0), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    orc = OraclePipeline(C.MINI_UNET, C.MINI_CONTROLNET, _cpu(wu), _cpu(wc), _cpu(wv))
    return eng, orc, text


@pytest.mark.parametrize("H,W,steps,cn", [(128, 128, 4, True), (96, 160, 2, True), (128, 128, 1, False),
                                          (120, 72, 2, True), (256, 256, 4, True)])
def test_mini_pipeline_matches_oracle(mini_setup, H, W, steps, cn):
    eng, orc, text = mini_setup
    eng.prepare(H, W, steps, 0.6, controlnet_scale=1.5, use_controlnet=cn)
    r0, r1, mad, psnr, _ = _compare(eng, orc, _frame(H, W), text, H, W, steps, cn)
    assert r0 <= 5e-3 and r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (r0, r1, mad, psnr)


@pytest.mark.parametrize("kind", ["black", "white", "flat_gray", "one_bright_pixel"])
def test_degenerate_frames_match_oracle(mini_setup, kind):
    """Edge cases of the reference's Sobel (canny_gpu.py:39-42): an all-black / constant frame has gradient 0 everywhere,
    `mag / mag.max()` is 0/0 = NaN, both thresholds are false for NaN and `ToPILImage` turns it into byte 0 -- the control
    image is black and nothing non-finite may reach the networks; a single bright pixel makes its neighbours the maximum."""
    eng, orc, text = mini_setup
    H = W = 128
    f = np.zeros((H, W, 3), dtype=np.uint8)
    if kind == "white":
        f[:] = 255
    elif kind == "flat_gray":
        f[:] = 117
    elif kind == "one_bright_pixel":
        f[40, 77] = 255
    eng.prepare(H, W, 2, 0.6, controlnet_scale=1.5, use_controlnet=True)
    r0, r1, mad, psnr, got = _compare(eng, orc, f, text, H, W, 2, True)
    assert torch.isfinite(eng.buffers["denoised"].float()).all() and torch.isfinite(eng.buffers["control"].float()).all()
    if kind != "one_bright_pixel":  # (the border of a constant non-black frame does have a gradient: zero padding)
        pass
    if kind == "black":
        assert float(eng.buffers["control"].float().abs().max()) == 0.0
    assert r0 <= 5e-3 and r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (kind, r0, r1, mad, psnr)


@pytest.mark.parametrize("strength,steps", [(1.0, 4), (0.05, 4), (0.6, 12), (0.98, 1)])
def test_schedule_extremes_match_oracle(mini_setup, strength, steps):
    """The client's option ranges (index.tsx:531-546: strength 0.05-1, steps 1-12): strength 1.0 starts at t = 999,
    strength 0.05 yields FEWER timesteps than `steps` ([39, 19]), 12 steps is the UI's maximum, 1 step draws no step noise."""
    eng, orc, text = mini_setup
    H, W = 96, 160
    plan = eng.prepare(H, W, steps, strength, controlnet_scale=1.0, use_controlnet=True)
    f = _frame(H, W, seed=17)
    got = eng.infer_u8(f)
    ref = np.asarray(orc.infer(Image.fromarray(f, "RGB"), text[None].float(), height=H, width=W, strength=strength, steps=steps,
                               seed=23, controlnet_scale=1.0, use_controlnet=True))
    assert plan["timesteps"] == orc.sched.timesteps.tolist()
    if (strength, steps) == (0.05, 4):
        assert plan["timesteps"] == [39, 19]
    if strength == 1.0:
        assert plan["timesteps"][0] == 999
    assert np.abs(got.astype(int) - ref.astype(int)).mean() <= 1.5 and _psnr(got, ref) >= 38.0


def test_graph_replay_is_deterministic_and_equals_eager(mini_setup):
    eng, orc, text = mini_setup
    f = _frame(128, 128, seed=3)
    eng.prepare(128, 128, 4, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=True)
    a = eng.infer_u8(f)
    b = eng.infer_u8(f)
    eng.prepare(128, 128, 4, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=False)
    c = eng.infer_u8(f)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    # a different frame changes the result; re-running the first frame restores it (no state leaks across frames)
    d = eng.infer_u8(_frame(128, 128, seed=4))
    assert not np.array_equal(a, d)
    assert np.array_equal(a, eng.infer_u8(f))


def test_option_sweep_with_one_capture_matches_fresh_prepares(mini_setup):
    """`controlnet_scale` and `strength` must not cost a re-capture.  One hipGraph; ten scales and three
    strengths through `update_options` (device constants the graph reads); every result bit-identical to a fresh
    `prepare` with those options, on the parent engine and on a slot that shares its constants."""
    eng, orc, text = mini_setup
    f = _frame(128, 128, seed=6)
    eng.prepare(128, 128, 4, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=True)
    graph = eng.graph
    slot = eng.make_slot()
    slot.prepare(128, 128, 4, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=True)
    sweep = [(0.6, s) for s in (0.05, 0.3, 0.55, 0.8, 1.05, 1.3, 1.75, 2.0, 2.5, 3.0)] + [(0.5, 1.0), (0.72, 2.0), (0.9, 0.4)]
    got, got_slot = [], []
    for strength, scale in sweep:
        assert eng.update_options(strength, scale)
        assert eng.graph is graph  # same captured graph
        got.append(eng.infer_u8(f).copy())
        got_slot.append(slot.infer_u8(f).copy())
    assert not eng.update_options(0.03, 1.0)  # a 1-step schedule is another program
    for (strength, scale), a, b in zip(sweep, got, got_slot):
        eng.prepare(128, 128, 4, strength, controlnet_scale=scale, use_controlnet=True, use_graph=True)
        ref = eng.infer_u8(f)
        assert np.array_equal(a, ref) and np.array_equal(b, ref), (strength, scale)
    assert len({g.tobytes() for g in got}) == len(sweep)  # and the options do change the image


def test_two_frames_in_flight_equal_sequential_results(mini_setup):
    """Slots share weights/constants but nothing mutable: concurrent frames are bit-identical to sequential ones."""
    eng, orc, text = mini_setup
    eng.prepare(128, 128, 4, 0.6, controlnet_scale=1.0, use_controlnet=True)
    frames = [_frame(128, 128, seed=s) for s in (11, 12, 13, 14)]
    seq = [eng.infer_u8(f).copy() for f in frames]
    slot = eng.make_slot()
    slot.prepare(128, 128, 4, 0.6, controlnet_scale=1.0, use_controlnet=True)
    engines = [eng, slot]
    for rep in range(3):
        outs = []
        for pair in ((0, 1), (2, 3)):
            for e, k in zip(engines, pair):
                e.ops.upload(e.frame_u8, torch.from_numpy(frames[k]))
                e.launch()  # both graphs are now in flight on different streams
            for e in engines:
                outs.append(e.ops.download(e.out_u8).numpy().copy())
        for got, ref in zip(outs, seq):
            assert np.array_equal(got, ref)


@pytest.mark.parametrize("H,W,cn", [(128, 128, True), (120, 72, True), (96, 160, False)])
def test_batched_frames_match_oracle_and_single_frames(mini_setup, H, W, cn):
    """prepare(batch=3): three frames per launch.  Every frame matches the oracle like a lone frame does, and stays
    within a few LSB of its single-frame result (different tiling / split-K => different fp32 summation order)."""
    eng, orc, text = mini_setup
    frames = np.stack([_frame(H, W, seed=s) for s in (21, 22, 23)])
    eng.prepare(H, W, 2, 0.6, controlnet_scale=1.5, use_controlnet=cn)
    single = np.stack([eng.infer_u8(f) for f in frames])
    eng.prepare(H, W, 2, 0.6, controlnet_scale=1.5, use_controlnet=cn, batch=3)
    got = eng.infer_u8(frames)
    assert np.array_equal(got, eng.infer_u8(frames))  # deterministic replay
    h0, w0 = H // 8, W // 8
    den = eng.buffers["denoised"][:, :4].float().cpu().reshape(3, h0, w0, 4).permute(0, 3, 1, 2)
    for b in range(3):
        ref = np.asarray(orc.infer(Image.fromarray(frames[b], "RGB"), text[None].float(), height=H, width=W, strength=0.6,
                                   steps=2, seed=23, controlnet_scale=1.5, use_controlnet=cn, keep_trace=True))
        ref_den = orc.trace["denoised"][-1][0]
        r1 = float((den[b] - ref_den).norm() / ref_den.norm())
        mad = float(np.abs(got[b].astype(int) - ref.astype(int)).mean())
        assert r1 <= 2e-2 and mad <= 1.5 and _psnr(got[b], ref) >= 38.0, (b, r1, mad)
        assert np.abs(got[b].astype(int) - single[b].astype(int)).mean() < 0.5
    eng.prepare(H, W, 2, 0.6, controlnet_scale=1.5, use_controlnet=cn, batch=1)


@pytest.fixture(scope="module")
def sd15_setup():
    from oracle.pipeline import OraclePipeline
    from videosd_amd import config as C
    from videosd_amd.engine import Engine
    from videosd_amd.ops import HipOps

    wu, wc, wv, text = _build(C.SD15_UNET, C.SD15_CONTROLNET)
    eng = Engine(HipOps(0), C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    orc = OraclePipeline(C.SD15_UNET, C.SD15_CONTROLNET, _cpu(wu), _cpu(wc), _cpu(wv))
    return eng, orc, text


def test_sd15_width_pipeline_matches_oracle(sd15_setup):
    """Full SD1.5 channel widths / head dims (40, 80, 160) and the real ControlNet tower, small frame."""
    eng, orc, text = sd15_setup
    H, W, steps = 128, 192, 2
    eng.prepare(H, W, steps, 0.6, controlnet_scale=1.0, use_controlnet=True)
    r0, r1, mad, psnr, _ = _compare(eng, orc, _frame(H, W), text, H, W, steps, True, cn_scale=1.0)
    assert r0 <= 5e-3 and r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (r0, r1, mad, psnr)


def test_baseline_config1_256_one_step_matches_oracle(sd15_setup):
    """BASELINE.json configs[0]: SD1.5 img2img 256x256, 1 LCM step (the single-step schedule draws no step noise)."""
    eng, orc, text = sd15_setup
    eng.prepare(256, 256, 1, 0.6, controlnet_scale=1.0, use_controlnet=True)
    assert eng.plan["timesteps"] == [599]
    r0, r1, mad, psnr, _ = _compare(eng, orc, _frame(256, 256, seed=5), text, 256, 256, 1, True, cn_scale=1.0)
    assert r0 <= 5e-3 and r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (r0, r1, mad, psnr)


@pytest.mark.parametrize("H,W,steps,scale", [(512, 512, 4, 1.0), (768, 768, 8, 2.0), (432, 768, 4, 2.0)])
def test_full_size_properties(sd15_setup, H, W, steps, scale):
    """Size-independent properties at BASELINE configs[1] (512x512, 4 steps), configs[4] (768x768, 8 steps,
    ControlNet scale 2) and the UI's live 768x432 frame (latent 96x54: upsample-to-skip-size path): finite,
    deterministic across replays, graph replay == eager run, frames do not leak into each other."""
    eng, orc, text = sd15_setup
    eng.prepare(H, W, steps, 0.6, controlnet_scale=scale, use_controlnet=True)
    assert len(eng.plan["timesteps"]) == steps
    f, g = _frame(H, W, seed=9), _frame(H, W, seed=10)
    a = eng.infer_u8(f)
    b = eng.infer_u8(g)
    assert np.array_equal(a, eng.infer_u8(f)) and not np.array_equal(a, b)
    assert torch.isfinite(eng.buffers["denoised"].float()).all() and a.std() > 1.0
    eng.prepare(H, W, steps, 0.6, controlnet_scale=scale, use_controlnet=True, use_graph=False)
    assert np.array_equal(a, eng.infer_u8(f))


def test_baseline_config2_512_four_step_matches_oracle(sd15_setup):
    """BASELINE.json configs[1] at FULL size against the oracle (lcm_controlnet.py:532-611): 512x512, 4 LCM steps
    [599,459,319,179], ControlNet on -- the exact bench workload, so the deep-K / large-M tile choices, split-K and the
    halo-patch convs that only occur at full size meet the oracle end to end.  One frame alone, then the batched plans the
    bench runs (3 frames per launch: round 1 / mid round 2; 5: the default now, with the fused tails on 80-token tiles): frame
    0 is the same frame (compared with the same oracle image), the others are compared with their single-frame results
    (other tiles / split-K => other fp32 summation order => a fraction of an LSB)."""
    eng, orc, text = sd15_setup
    H = W = 512
    eng.prepare(H, W, 4, 0.6, controlnet_scale=1.0, use_controlnet=True)
    assert eng.plan["timesteps"] == [599, 459, 319, 179]
    frames = np.stack([_frame(H, W, seed=s) for s in (31, 32, 33, 34, 35)])
    r0, r1, mad, psnr, got = _compare(eng, orc, frames[0], text, H, W, 4, True, cn_scale=1.0)
    assert r0 <= 5e-3 and r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (r0, r1, mad, psnr)
    single = [got] + [eng.infer_u8(f).copy() for f in frames[1:]]
    ref_den = orc.trace["denoised"][-1][0]
    for nb in (3, 5):
        eng.prepare(H, W, 4, 0.6, controlnet_scale=1.0, use_controlnet=True, batch=nb)
        out = eng.infer_u8(frames[:nb])
        den = eng.buffers["denoised"][:, :4].float().cpu().reshape(nb, H // 8, W // 8, 4).permute(0, 3, 1, 2)
        assert float((den[0] - ref_den).norm() / ref_den.norm()) <= 2e-2
        for b in range(nb):
            d = np.abs(out[b].astype(int) - single[b].astype(int))
            assert d.mean() < 0.5 and _psnr(out[b], single[b]) >= 42.0, (nb, b, d.mean())
    eng.prepare(H, W, 4, 0.6, controlnet_scale=1.0, use_controlnet=True, batch=1)


@pytest.mark.parametrize("case", ["stress512m", "stress512"])
def test_baseline_config2_on_range_stress_weights_matches_oracle(case):
    """BASELINE configs[1] (512x512, 4 steps, ControlNet) on the RANGE-STRESS weight sets (weights.synthesize(stress=1 / 2)): what
    trained SD1.5 tensors look like and N(0, 1/fan_in) does not -- per-output-channel scales over one / two decades, 0.5 % outlier
    channels 8 x / 30 x larger, GroupNorm / LayerNorm gamma away from 1 with beta ~ 0.3 N, and (level 2) attention logits pushed to
    +-30.  Every parity number of this suite used to be on the plain set.
    The yardstick: on these weights fp16 STORAGE alone moves the network's output (the oracle with every layer output rounded to
    fp16, oracle.nets.EMULATE_FP16, against the fp32 oracle: stored beside the fp32 answer) -- by 27 % of the denoised latents on the
    level-2 set, where peaked softmax rows turn rounding differences into different attention; any fp16 pipeline, the reference's own
    included, moves that much there.  The HIP path must be finite and at most 1.5 x that far from the fp32 oracle (and, where the
    yardstick is below the plain set's tolerance, within the plain tolerance).  On level 2 this is an OVERFLOW / FINITENESS gate, not
    a closeness gate: the tight check on these weights is the one-forward test below, before four steps amplify
    anything."""
    from videosd_amd import config as C
    from videosd_amd import weights as W
    from videosd_amd.engine import Engine
    from videosd_amd.ops import HipOps

    c = GOLDEN_CASES[case]
    wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda", stress=c["stress"])
    wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda", stress=c["stress"])
    wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
    text = (torch.randn(77, C.SD15_UNET.cross_dim, generator=torch.Generator().manual_seed(c["text_seed"])) * 0.5).half()
    ops = HipOps(0)
    ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
    eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    H, W_ = c["H"], c["W"]
    eng.prepare(H, W_, c["steps"], c["strength"], controlnet_scale=c["cn_scale"], use_controlnet=True)
    frame = _frame(H, W_, seed=c["frame_seed"])
    r0, r1, mad, psnr, got = _compare_golden(eng, frame, H, W_, case)
    with np.load(GOLDEN_FULLSIZE) as z:
        ref = torch.from_numpy(z[case + "_denoised"]).float()
        emu = torch.from_numpy(z[case + "_fp16emu_denoised"]).float()
        mad_emu = float(np.abs(z[case + "_fp16emu_image_half"].astype(int) - z[case + "_image_half"].astype(int)).mean())
    r1_emu = float((emu - ref).norm() / ref.norm())
    den = eng.buffers["denoised"][:, :4].float()
    assert bool(torch.isfinite(den).all()), "non-finite latents on the stress weights"
    assert r0 <= 5e-3, r0
    assert r1 <= max(2e-2, 1.5 * r1_emu) and mad <= max(1.5, 1.5 * mad_emu), (case, r1, r1_emu, mad, mad_emu, psnr)
    # five frames per launch (the bench's plan, throughput-mode kernels): frame 0 is the same frame, as close to the one-frame
    # result as fp16 storage allows on this set
    eng.tune_for_lanes = True
    eng.prepare(H, W_, c["steps"], c["strength"], controlnet_scale=c["cn_scale"], use_controlnet=True, batch=5)
    out = eng.infer_u8(np.stack([frame] + [_frame(H, W_, seed=s) for s in (72, 73, 74, 75)]))
    d = np.abs(out[0].astype(int) - got.astype(int))
    assert d.mean() <= max(0.5, 1.5 * mad_emu), (case, d.mean(), mad_emu)


@pytest.mark.parametrize("stress", [1, 2])
def test_one_forward_on_range_stress_weights_against_the_live_oracle(stress):
    """the four-step level-2 case above can only be held to 1.5 x a 27 % yardstick -- four denoising steps of peaked
    attention amplify any rounding difference, so that gate proves "finite, and no worse than fp16 storage" and little else.  Here
    the error is measured BEFORE that amplification: ONE ControlNet + UNet forward (a one-step schedule, SD1.5 widths, 256 x 256) on
    the same range-stress weights, against the LIVE fp32 oracle, with the oracle's own fp16-storage emulation as the yardstick
    computed in the same test.  The HIP path's noise prediction and denoised latents must be within the plain tolerance, or -- where
    fp16 storage alone already exceeds it -- within 1.1 x what fp16 storage costs; the TAESD-encoded latents and the ControlNet's
    conditioning embedding (inputs of that forward) within the plain per-tensor tolerances."""
    from oracle.pipeline import OraclePipeline
    from videosd_amd import config as C
    from videosd_amd import weights as W
    from videosd_amd.engine import Engine
    from videosd_amd.ops import HipOps

    wu = W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda", stress=stress)
    wc = W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda", stress=stress)
    wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
    text = (torch.randn(77, C.SD15_UNET.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5).half()
    ops = HipOps(0)
    ops.load_tuning(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tuning_mi355x.json"))
    eng = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    H = W_ = 256
    eng.prepare(H, W_, 1, 0.6, controlnet_scale=1.0, use_controlnet=True)
    frame = _frame(H, W_, seed=31)
    eng.infer_u8(frame)
    h0, w0 = H // 8, W_ // 8
    nchw = lambda b: eng.buffers[b][:, :4].float().cpu().reshape(h0, w0, 4).permute(2, 0, 1)  # noqa: E731
    got = {"x0": nchw("x0"), "eps": nchw("eps"), "den": nchw("denoised")}
    assert all(bool(torch.isfinite(v).all()) for v in got.values())
    orc = OraclePipeline(C.SD15_UNET, C.SD15_CONTROLNET, _cpu(wu), _cpu(wc), _cpu(wv))
    res = {}
    for name, emu in (("fp32", False), ("emu", True)):
        orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=H, width=W_, strength=0.6, steps=1, seed=23,
                  controlnet_scale=1.0, use_controlnet=True, keep_trace=True, emulate_fp16=emu)
        res[name] = {"x0": orc.trace["init_latents"][0].clone(), "eps": orc.trace["eps"][0][0].clone(), "den": orc.trace["denoised"][0][0].clone()}
    rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
    r = {k: rel(got[k], res["fp32"][k]) for k in got}
    y = {k: rel(res["emu"][k], res["fp32"][k]) for k in got}
    print(f"stress {stress}: HIP vs fp32 {r}, fp16-storage yardstick {y}")
    assert r["x0"] <= 5e-3, r
    for k in ("eps", "den"):
        assert r[k] <= max(2e-2, 1.1 * y[k]), (stress, k, r, y)


def test_side_stream_and_single_stream_sequences_give_the_same_bits(mini_setup):
    """Every engine holds two forms of its program (Engine.prepare / _capture): the ControlNet encoder on the lane's side stream
    (graphs + event edges) and everything on the lane's own stream (one graph), where the two encoders walk in lock step and their
    twin layers share a grid (ops.pair; round 5).  Same buffers, and every kernel sums in the same order in both: the frame must
    not depend on which of the two a launch takes -- nor on another lane running beside it."""
    eng, orc, text = mini_setup
    eng.overlap_controlnet = True
    eng.prepare(128, 128, 4, 0.6, controlnet_scale=1.5, use_controlnet=True, use_graph=True)  # (the fixture's engine may come from an eager test)
    assert eng.plan["edges"] == 2 * 4 and eng.plan["graphs"] == 1 + 3 * 4 and eng.graph_serial is not eng.graph
    assert eng.ops.seq_count(eng.graph_serial) == (1, 0)
    n2, k2 = eng.launches_by_kind()
    n1, k1 = eng.launches_by_kind(serial=True)
    assert eng.program_serial is not eng.program and n1 < 0.85 * n2 and k1.get("pair_conv", 0) > 20 and not any(k.startswith("pair") for k in k2)
    f = _frame(128, 128, seed=9)
    eng.overlap_launch = True
    a = eng.infer_u8(f)
    eng.overlap_launch = False
    b = eng.infer_u8(f)
    assert np.array_equal(a, b)
    other = eng.make_slot()          # lane 1, busy beside lane 0
    other.prepare(128, 128, 4, 0.6, controlnet_scale=1.5, use_controlnet=True)
    for ov in (True, False):
        other.submit_u8(_frame(128, 128, seed=10), overlap=ov)
        eng.submit_u8(f, overlap=ov)
        assert np.array_equal(eng.collect_u8(), a)
        other.collect_u8()
    eng.overlap_launch = True
    assert eng.ops.pool_check() < 1.5  # the four launch streams sit on four different command-processor pipes


def test_stored_and_live_oracle_comparisons_agree():
    """The full-size cases are compared with the oracle's STORED output (`_compare_golden`), everything else with the live
    oracle (`_compare`).  One small case goes through both in the same test -- same engine, same frame -- so that the two code
    paths cannot drift apart: the stored numbers must be the live numbers (up to the fixture's fp16 latents / half-resolution
    image).  CPU-generator weights: the fixture's `mini64` arrays are reproducible in any container (--only mini64)."""
    from oracle.pipeline import OraclePipeline
    from videosd_amd import config as C
    from videosd_amd import weights as W
    from videosd_amd.engine import Engine
    from videosd_amd.ops import HipOps

    c = GOLDEN_CASES["mini64"]
    wu, wc, wv = (W.synthesize(W.unet_spec(C.MINI_UNET), "unet."), W.synthesize(W.controlnet_spec(C.MINI_CONTROLNET), "cn."),
                  W.synthesize(W.taesd_spec(C.TAESD), "vae."))
    text = (torch.randn(77, C.MINI_UNET.cross_dim, generator=torch.Generator().manual_seed(c["text_seed"])) * 0.5).half()
    dev = lambda w: {k: v.cuda() for k, v in w.items()}  # noqa: E731
    eng = Engine(HipOps(0), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, dev(wu), dev(wc), dev(wv))
    eng.set_text_embeds(text)
    eng.prepare(c["H"], c["W"], c["steps"], c["strength"], controlnet_scale=c["cn_scale"], use_controlnet=True)
    frame = _frame(c["H"], c["W"], seed=c["frame_seed"])
    orc = OraclePipeline(C.MINI_UNET, C.MINI_CONTROLNET, wu, wc, wv)
    live = _compare(eng, orc, frame, text, c["H"], c["W"], c["steps"], True, cn_scale=c["cn_scale"])
    stored = _compare_golden(eng, frame, c["H"], c["W"], "mini64")
    assert np.array_equal(live[4], stored[4])                                          # the same engine frame both times
    assert live[0] <= 5e-3 and live[1] <= 2e-2 and live[2] <= 1.5 and live[3] >= 38.0, live[:4]
    assert abs(live[0] - stored[0]) < 2e-4 and abs(live[1] - stored[1]) < 2e-4, (live[:2], stored[:2])  # fp16 storage of the latents
    assert abs(live[2] - stored[2]) < 0.1 and abs(live[3] - stored[3]) < 1.0, (live[2:4], stored[2:4])  # every second row / column
    # and the stored image IS the live oracle's image
    want = np.asarray(orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=c["H"], width=c["W"], strength=c["strength"],
                                steps=c["steps"], seed=23, controlnet_scale=c["cn_scale"], use_controlnet=True))
    with np.load(GOLDEN_FULLSIZE) as z:
        assert np.array_equal(z["mini64_image_half"], want[::2, ::2])


@pytest.mark.slow
def test_baseline_config5_768_eight_step_scale2_matches_oracle(sd15_setup):
    """BASELINE.json configs[4] at full size: 768x768, 8 steps, ControlNet scale 2, against the oracle's frame of exactly
    these inputs (tests/golden/fullsize_oracle.npz, written on the GPU box by scripts/make_fullsize_golden.py; with
    VSD_LIVE_ORACLE=1 the oracle runs here: about 3 minutes)."""
    eng, orc, text = sd15_setup
    c = GOLDEN_CASES["config5"]
    H, W = c["H"], c["W"]
    eng.prepare(H, W, c["steps"], c["strength"], controlnet_scale=c["cn_scale"], use_controlnet=True)
    assert (H, W) == (768, 768) and eng.plan["timesteps"] == [599, 539, 479, 419, 359, 299, 239, 179]
    if os.environ.get("VSD_LIVE_ORACLE") == "1":
        r0, r1, mad, psnr, _ = _compare(eng, orc, _frame(H, W, seed=c["frame_seed"]), text, H, W, c["steps"], True, cn_scale=c["cn_scale"])
    else:
        r0, r1, mad, psnr, _ = _compare_golden(eng, _frame(H, W, seed=c["frame_seed"]), H, W, "config5")
    assert r0 <= 5e-3 and r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (r0, r1, mad, psnr)


@pytest.mark.slow
def test_reference_only_mode_512_four_step_matches_oracle(sd15_setup):
    """SURVEY 8f-4 at FULL size (tested at 128x192 and 64x64 only): the reference-only program
    (lcm_reference_pipeline.py:498-794, 855-890: per step a WRITE pass over the noised reference latents banks self-attention
    K / V and block statistics, the READ pass attends over [x ; bank] -- 8192 keys at the 64x64 level -- and applies AdaIN) at
    512x512, 4 steps, against the oracle restatement; and what it costs (the mode doubles the UNet work)."""
    import time

    eng, orc, text = sd15_setup
    c = GOLDEN_CASES["ref512"]
    H, W = c["H"], c["W"]
    eng.prepare(H, W, c["steps"], c["strength"], use_controlnet=False, ref_mode=True)
    frame, refimg = _frame(H, W, seed=c["frame_seed"]), _frame(H, W, seed=c["ref_seed"])
    eng.ops.upload(eng.ref_u8, torch.from_numpy(refimg))
    got = eng.infer_u8(frame)
    den = eng.buffers["denoised"][:, :4].float().cpu().reshape(H // 8, W // 8, 4).permute(2, 0, 1)
    if os.environ.get("VSD_LIVE_ORACLE") == "1":
        want = np.asarray(orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=H, width=W, strength=0.6, steps=4,
                                    seed=23, ref_image=Image.fromarray(refimg, "RGB"), keep_trace=True))
        ref_den, half = orc.trace["denoised"][-1][0], got
    else:  # the oracle's output for exactly these inputs, stored by scripts/make_fullsize_golden.py (50 s of oracle otherwise)
        with np.load(GOLDEN_FULLSIZE) as z:
            want, ref_den = z["ref512_image_half"], torch.from_numpy(z["ref512_denoised"]).float()
        half = got[::2, ::2]
    r1 = float((den - ref_den).norm() / ref_den.norm())
    mad = float(np.abs(half.astype(int) - want.astype(int)).mean())
    assert r1 <= 2e-2 and mad <= 1.5 and _psnr(half, want) >= 38.0, (r1, mad, _psnr(half, want))
    assert np.array_equal(got, eng.infer_u8(frame))  # deterministic replay
    lat = []
    for _ in range(6):
        t0 = time.perf_counter()
        eng.infer_u8(frame)
        lat.append((time.perf_counter() - t0) * 1e3)
    print(f"reference-only 512x512 4-step: {sorted(lat)[3]:.1f} ms/frame (host u8 in -> host u8 out, one frame per launch)")
    eng.prepare(H, W, 4, 0.6, controlnet_scale=1.0, use_controlnet=True)  # (leave the shared engine in its usual state)


@pytest.mark.parametrize("cfg_name", ["MINI_CLIP", "CLIP_L"])
def test_clip_text_encoder_matches_oracle(cfg_name):
    """CLIP text encoder on the HIP kernels vs the oracle restatement (itself pinned against transformers)."""
    from oracle import nets
    from videosd_amd import config as C
    from videosd_amd import weights as W
    from videosd_amd.clip import ClipTextEncoder
    from videosd_amd.ops import HipOps

    cfg = getattr(C, cfg_name)
    w = W.synthesize(W.clip_spec(cfg), "clip.", device="cuda")
    enc = ClipTextEncoder(HipOps(0), cfg, w)
    ids = torch.randint(0, cfg.vocab, (cfg.max_len,), generator=torch.Generator().manual_seed(3))
    got = enc.encode_ids(ids).float().cpu()
    ref = nets.clip_text_forward({k: v.cpu() for k, v in w.items()}, cfg, ids[None])[0]
    rel = float((got - ref).norm() / ref.norm())
    assert torch.isfinite(got).all() and rel <= 1e-2, rel


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 3, 5])
def test_soak_two_launches_in_flight_full_size_bit_identical(batch):
    """The bench configuration at full size (512x512, 4 steps, ControlNet), two launches in flight, replayed: every result
    bit-identical to the sequential one (scripts/soak.py; this is the test that caught the per-pixel gather Sobel)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "vsd_soak", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    assert soak.run(n={1: 60, 3: 30}.get(batch, 20), batch=batch, verbose=False) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("batch,lanes,side", [(5, 4, False), (1, 4, False), (1, 2, True), (3, 2, True)])
def test_soak_four_launch_streams_busy_full_size_bit_identical(batch, lanes, side):
    """Round 4's concurrency machinery under load, at full size: four launch lanes in flight (the bench's 5 x 4 operating point
    with its throughput-mode kernel choices; four one-frame launches), and two lanes whose ControlNet encoders run on the
    lanes' side streams (launch sequences of 13 graphs + 8 event edges each: all four launch streams busy).  Every result
    bit-identical to the sequential result of the same input -- no interference through shared scratch, split-K tickets,
    prompt constants or the streams themselves."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "vsd_soak", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    assert soak.run(n={1: 64}.get(batch, 24), batch=batch, verbose=False, lanes=lanes, side=side) == 0


def test_no_kernel_reads_uninitialised_memory():
    """VSD_POISON=1 fills everything the engine allocates "uninitialised" with NaN bytes: the live program and the
    reference-only program still match the oracle, at sizes whose token counts are ragged against the 64-key attention
    tile at every level.  (Found this way: the reference-only WRITE pass read the masked tail of its last key tile from
    the next V^T row, and 0 x NaN poisoned the banked statistics whenever the allocator handed out dirty memory.)"""
    import json
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, VSD_POISON="1")
    res = subprocess.run([sys.executable, os.path.join(here, "poison_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    diffs = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert set(diffs) == {"live", "reference_only", "reference_only_64"}
    assert all(v < 1.5 for v in diffs.values()), diffs


@pytest.mark.gpu
def test_whole_program_with_every_buffer_between_unmapped_pages():
    """scripts/guard_page_engine.py: SD1.5 + ControlNet and the mini SDXL topology at ragged / degenerate frame sizes, 1 and 3
    frames per launch, with every device buffer of the engine (packed weights, each activation tensor, workspaces, I/O) ending
    at or starting after an unmapped page (`HipOps.allocator` hook + HIP's virtual-memory API): an out-of-bounds access anywhere
    in the recorded program faults; every frame equals the ordinary engine's bit for bit."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "guard_page_engine.py"), "8x8", "24x40", "104x88", "200x136"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "guard page engine run passed" in r.stdout, (r.stdout[-800:], r.stderr[-1500:])
