"""BASELINE.json configs[3] (SDXL 1024x1024 LCM 4-step, larger UNet) on the MI355X, through the C-ABI.

SDXL is not in the reference: the per-frame loop is the reference's (lcm_controlnet.py:379-618,
TAESD encode -> LCM steps -> TAESD decode) with the SDXL-base UNet (3 levels, 2 / 10 BasicTransformerBlocks per
Transformer2D, Linear proj_in/out, head size 64, cross_dim 2048, text_time added conditioning), no ControlNet.
Same tolerances as tests/test_pipeline_gpu.py."""
import os
import time

import numpy as np
import pytest
import torch
from PIL import Image

from test_pipeline_gpu import _compare_golden, _cpu, _frame, _psnr

pytestmark = pytest.mark.gpu


def _setup(cfg, prefix):
    from oracle.pipeline import OraclePipeline
    from videosd_amd import config as C
    from videosd_amd import weights as W
    from videosd_amd.engine import Engine
    from videosd_amd.ops import HipOps

    wu = W.synthesize(W.unet_spec(cfg), prefix, device="cuda")
    wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")
    g = torch.Generator().manual_seed(11)
    text = (torch.randn(77, cfg.cross_dim, generator=g) * 0.5).half()
    pooled = (torch.randn(cfg.add_pooled_dim, generator=g) * 0.5).half()
    eng = Engine(HipOps(0), cfg, None, C.TAESD, wu, None, wv)
    eng.set_text_embeds(text)
    orc = OraclePipeline(cfg, None, _cpu(wu), None, _cpu(wv))
    return eng, orc, text, pooled


def _compare(eng, orc, text, pooled, H, W, steps):
    frame = _frame(H, W, seed=2)
    eng.set_added_cond(pooled, (H, W, 0, 0, H, W))
    eng.prepare(H, W, steps, 0.6, use_controlnet=False)
    got = eng.infer_u8(frame)
    ref = np.asarray(orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=H, width=W, strength=0.6,
                               steps=steps, seed=23, use_controlnet=False, keep_trace=True, pooled=pooled))
    h0, w0 = H // 8, W // 8
    den = eng.buffers["denoised"][:, :4].float().cpu().reshape(h0, w0, 4).permute(2, 0, 1)
    ref_den = orc.trace["denoised"][-1][0]
    r1 = float((den - ref_den).norm() / ref_den.norm())
    mad = float(np.abs(got.astype(int) - ref.astype(int)).mean())
    return r1, mad, _psnr(got, ref)


@pytest.fixture(scope="module")
def mini_xl():
    from videosd_amd import config as C

    return _setup(C.MINI_SDXL_UNET, "xl.")


@pytest.mark.parametrize("H,W,steps", [(128, 128, 4), (96, 160, 2), (120, 72, 1), (256, 256, 2)])
def test_mini_sdxl_pipeline_matches_oracle(mini_xl, H, W, steps):
    eng, orc, text, pooled = mini_xl
    r1, mad, psnr = _compare(eng, orc, text, pooled, H, W, steps)
    assert r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (r1, mad, psnr)


@pytest.fixture(scope="module")
def full_xl():
    from videosd_amd import config as C

    return _setup(C.SDXL_UNET, "sdxl.")


def test_sdxl_width_pipeline_matches_oracle(full_xl):
    """Full SDXL channel widths / depths (2.57 G parameters), small frame so the CPU oracle finishes in seconds."""
    eng, orc, text, pooled = full_xl
    r1, mad, psnr = _compare(eng, orc, text, pooled, 128, 192, 2)
    assert r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (r1, mad, psnr)


def test_sdxl_1024_properties(full_xl):
    """BASELINE.json configs[3] at full size: finite, deterministic across replays, graph replay == eager run, frames
    do not leak into each other."""
    eng, orc, text, pooled = full_xl
    H = W = 1024
    eng.set_added_cond(pooled, (H, W, 0, 0, H, W))
    eng.prepare(H, W, 4, 0.6, use_controlnet=False)
    assert eng.plan["timesteps"] == [599, 459, 319, 179]
    f, g = _frame(H, W, seed=9), _frame(H, W, seed=10)
    a = eng.infer_u8(f)
    b = eng.infer_u8(g)
    t0 = time.perf_counter()
    a2 = eng.infer_u8(f)
    print(f"SDXL 1024x1024 4-step: {(time.perf_counter() - t0) * 1e3:.1f} ms/frame (host u8 in -> host u8 out)")
    assert np.array_equal(a, a2) and not np.array_equal(a, b)
    assert torch.isfinite(eng.buffers["denoised"].float()).all() and a.std() > 1.0
    eng.prepare(H, W, 4, 0.6, use_controlnet=False, use_graph=False)
    assert np.array_equal(a, eng.infer_u8(f))


@pytest.mark.slow
def test_sdxl_1024_four_step_matches_oracle(full_xl):
    """BASELINE.json configs[3] at FULL size against the oracle (properties only until now): 1024x1024, 4 LCM steps,
    the 10-deep transformer stacks at 32x32 and the two-block stacks at 64x64 with the tiles / split-K full size selects.
    (tests/golden/fullsize_oracle.npz; VSD_LIVE_ORACLE=1 runs the oracle here: 3 minutes, 27 TFLOP in fp32.)"""
    eng, orc, text, pooled = full_xl
    if os.environ.get("VSD_LIVE_ORACLE") == "1":
        r1, mad, psnr = _compare(eng, orc, text, pooled, 1024, 1024, 4)
    else:  # the oracle's frame of exactly these inputs, stored by scripts/make_fullsize_golden.py
        eng.set_added_cond(pooled, (1024, 1024, 0, 0, 1024, 1024))
        eng.prepare(1024, 1024, 4, 0.6, use_controlnet=False)
        _, r1, mad, psnr, _ = _compare_golden(eng, _frame(1024, 1024, seed=2), 1024, 1024, "sdxl1024")
    assert r1 <= 2e-2 and mad <= 1.5 and psnr >= 38.0, (r1, mad, psnr)
