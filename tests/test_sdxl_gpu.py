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


@pytest.mark.parametrize("size", ["mini", "full"])
def test_sdxl_text_towers_match_oracle(size):
    """Both SDXL text towers on the HIP kernels (CLIP-L quick-GELU; OpenCLIP bigG: 32 layers, erf GELU = VSD_ACT_GELU, text
    projection) against the oracle restatement, itself pinned against transformers (tests/test_oracle_text_encoders.py):
    prompt_embeds = cat(hidden_states[-2]) and pooled = tower 2's text_embeds."""
    from oracle import text_encoders as T
    from videosd_amd import clip as K
    from videosd_amd import weights as W
    from videosd_amd.ops import HipOps

    c1 = K.SDXL_CLIP_L if size == "full" else K.TextTowerConfig(vocab=1000, width=128, heads=2, layers=2, mlp=512)
    c2 = K.SDXL_CLIP_G if size == "full" else K.MINI_CLIP_G
    w1 = W.synthesize(K.text_tower_spec(c1), "t1.", device="cuda")
    w2 = W.synthesize(K.text_tower_spec(c2), "t2.", device="cuda")
    ops = HipOps(0)
    enc = K.SdxlTextEncoders(K.ClipTextEncoder(ops, c1, w1), K.ClipTextEncoder(ops, c2, w2))
    g = torch.Generator().manual_seed(3)
    ids1 = torch.randint(1, c1.vocab - 1, (77,), generator=g)
    ids2 = ids1.clone() if c1.vocab == c2.vocab else torch.randint(1, c2.vocab - 1, (77,), generator=g)
    ids1[12:], ids2[12], ids2[13:] = c1.vocab - 1, c2.vocab - 1, 0   # end-of-text at 12; tower 2 pads with id 0
    emb, pooled = enc.encode_ids(ids1, ids2)
    want_e, want_p = T.sdxl_prompt_embeds(_cpu(w1), c1, ids1[None], _cpu(w2), c2, ids2[None])
    emb, pooled = emb.float().cpu(), pooled.float().cpu()
    assert emb.shape == (77, c1.width + c2.width) and pooled.shape == (c2.proj,)
    assert torch.isfinite(emb).all() and torch.isfinite(pooled).all()
    for got, want, tol in ((emb[:, :c1.width], want_e[0][:, :c1.width], 1e-2), (emb[:, c1.width:], want_e[0][:, c1.width:], 1e-2),
                           (pooled, want_p[0], 1e-2)):
        rel = float((got - want).norm() / want.norm())
        assert rel <= tol, rel
    # the pooled row is the end-of-text row: another token after it changes nothing before it (causal), the pooled vector stays
    ids2b = ids2.clone()
    ids2b[20] = 5
    assert torch.equal(enc.t2.text_embeds(ids2b).float().cpu(), pooled)


def test_sdxl_snapshot_with_both_text_encoders_feeds_the_engine(tmp_path, monkeypatch):
    """An SDXL snapshot directory (<model>/text_encoder, text_encoder_2, tokenizer, tokenizer_2 as diffusers lays them out): the
    drop-in class tokenises with both tokenizers (tokenizer_2 pads with "!" = id 0), runs both towers on the HIP kernels and
    conditions the engine on their outputs instead of the stand-ins."""
    from safetensors.torch import save_file

    from oracle import text_encoders as T
    from test_checkpoint_gpu import write_toy_clip_tokenizer
    from videosd_amd import clip as K
    from videosd_amd import weights as W
    from videosd_amd.pipeline import VideoSDPipeline

    root = os.path.join(str(tmp_path), "lcm-sdxl-snapshot")
    for sub in ("text_encoder", "text_encoder_2", "tokenizer", "tokenizer_2"):
        os.makedirs(os.path.join(root, sub))
    tow = {}
    for sub, cfg, prefix in (("text_encoder", K.SDXL_CLIP_L, "t1."), ("text_encoder_2", K.SDXL_CLIP_G, "t2.")):
        tow[sub] = {k: v.cpu().contiguous() for k, v in W.synthesize(K.text_tower_spec(cfg), prefix, device="cuda").items()}
        save_file(tow[sub], os.path.join(root, sub, "model.safetensors"))
    for sub in ("tokenizer", "tokenizer_2"):
        write_toy_clip_tokenizer(os.path.join(root, sub))
    monkeypatch.delenv("VSD_WEIGHTS", raising=False)
    p = VideoSDPipeline(model=root, controlnet="none", tuning_mode="table")
    assert p.is_xl and p.text_encoder is not None and p.text_encoder.has_tokenizer
    assert p.weight_sources["text_encoder_2"].endswith(os.path.join("text_encoder_2", "model.safetensors"))
    prompt = "pixar, cg"
    i1, i2 = p.text_encoder.t1.tokenize(prompt), p.text_encoder.t2.tokenize(prompt)
    n = int((i1 != i1[-1]).sum())
    assert torch.equal(i1[:n + 1], i2[:n + 1]) and int(i1[-1]) == int(i1.max()) and int(i2[-1]) == 0   # same tokens, other padding
    emb, pooled = p.encode_prompt(prompt), p.encode_pooled(prompt)
    want_e, want_p = T.sdxl_prompt_embeds(tow["text_encoder"], K.SDXL_CLIP_L, i1[None], tow["text_encoder_2"], K.SDXL_CLIP_G, i2[None])
    assert emb.shape == (77, 2048) and pooled.shape == (1280,)
    assert float((emb.float().cpu() - want_e[0]).norm() / want_e.norm()) < 1e-2
    assert float((pooled.float().cpu() - want_p[0]).norm() / want_p.norm()) < 1e-2
    # the frame is conditioned on them: the same pipeline given the stand-in embeddings of that text renders another frame
    img = Image.fromarray(_frame(200, 300, seed=9), "RGB")
    opts = dict(prompt=prompt, height=128, width=192, strength=0.6, steps=2)
    a = np.asarray(p.infer(img, **opts))
    assert np.array_equal(np.asarray(p.infer(img, **opts)), a)
    q = VideoSDPipeline(model="latent-consistency/lcm-sdxl", controlnet="none", tuning_mode="table")
    assert q.text_encoder is None and "stand-in" in q.weight_sources["text_encoder_2"]
    b = np.asarray(q.infer(img, **opts))
    assert a.shape == b.shape == (128, 192, 3) and np.abs(a.astype(int) - b.astype(int)).mean() > 1.0
    # a second tower without its projection matrix is refused by name
    bad = dict(tow["text_encoder_2"])
    del bad["text_projection.weight"]
    save_file(bad, os.path.join(root, "text_encoder_2", "model.safetensors"))
    with pytest.raises(KeyError, match="text_projection.weight"):
        VideoSDPipeline(model=root, controlnet="none", tuning_mode="table")
