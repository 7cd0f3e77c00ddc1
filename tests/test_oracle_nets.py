"""Oracle network restatements: parameter counts, CLIP pinned against transformers, mini pipeline runs."""
import numpy as np
import torch
from PIL import Image

from oracle import nets
from oracle.pipeline import OraclePipeline, center_crop_resize, sobel_edges
from videosd_amd import config as C
from videosd_amd import weights as W


def test_param_counts_match_published_sizes():
    assert round(W.count_params(W.unet_spec(C.SD15_UNET)) / 1e6, 2) == 859.60
    assert round(W.count_params(W.controlnet_spec(C.SD15_CONTROLNET)) / 1e6, 2) == 361.28
    s = W.taesd_spec(C.TAESD)
    assert W.count_params([x for x in s if x[0].startswith("encoder")]) == 1_222_532
    assert W.count_params([x for x in s if x[0].startswith("decoder")]) == 1_222_531
    assert round(W.count_params(W.clip_spec(C.CLIP_L)) / 1e6, 2) == 123.06
    # SDXL-base UNet (2 567 463 684) + the LCM guidance projection 256 -> 320 (81 920): BASELINE.json configs[3]
    assert W.count_params(W.unet_spec(C.SDXL_UNET)) == 2_567_463_684 + 81_920


def test_clip_restatement_matches_transformers():
    import transformers

    cfg = C.MINI_CLIP
    hf_cfg = transformers.CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.width, intermediate_size=cfg.mlp,
                                         num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                                         max_position_embeddings=cfg.max_len, hidden_act="quick_gelu",
                                         layer_norm_eps=cfg.eps, bos_token_id=0, eos_token_id=cfg.vocab - 1)
    m = transformers.CLIPTextModel(hf_cfg).eval().float()
    w = W.synthesize(W.clip_spec(cfg), "clip.", dtype=torch.float32)
    keys = set(m.state_dict().keys())
    sd = w if any(k.startswith("text_model.") for k in keys) else {k[len("text_model."):]: v for k, v in w.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    ids = torch.randint(0, cfg.vocab, (1, cfg.max_len), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = m(ids)[0]
        got = nets.clip_text_forward(w, cfg, ids)
    assert torch.allclose(ref, got, atol=2e-5, rtol=1e-4), float((ref - got).abs().max())


def test_crop_and_sobel_shapes():
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 256, (90, 160, 3), dtype=np.uint8), "RGB")
    out = center_crop_resize(img, 64, 48)
    assert out.size == (64, 48)
    e = sobel_edges(out)
    a = np.asarray(e)
    assert e.mode == "L" and a.shape == (48, 64) and a.max() == 255


def _mini_pipeline():
    wu = W.synthesize(W.unet_spec(C.MINI_UNET), "unet.")
    wc = W.synthesize(W.controlnet_spec(C.MINI_CONTROLNET), "cn.")
    wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.")
    return OraclePipeline(C.MINI_UNET, C.MINI_CONTROLNET, wu, wc, wv)


def test_mini_pipeline_runs_and_is_deterministic():
    p = _mini_pipeline()
    rng = np.random.default_rng(1)
    img = Image.fromarray(rng.integers(0, 256, (72, 96, 3), dtype=np.uint8), "RGB")
    text = torch.randn(1, 77, C.MINI_UNET.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5
    a = p.infer(img, text, height=48, width=72, strength=0.6, steps=2, seed=23, controlnet_scale=1.0)
    b = p.infer(img, text, height=48, width=72, strength=0.6, steps=2, seed=99, controlnet_scale=1.0)
    assert a.size == (72, 48)
    # the CPU RNG is reset to a fresh generator state on every frame, so `seed` does not matter
    assert np.array_equal(np.asarray(a), np.asarray(b))
    c = p.infer(img, text, height=48, width=72, strength=0.6, steps=2, seed=23, controlnet_scale=2.0)
    assert not np.array_equal(np.asarray(a), np.asarray(c))


def test_oracle_nets_against_diffusers_goldens():
    """UNet / ControlNet / TAESD forwards of oracle/nets.py against outputs of diffusers' own modules on the same seeded synthetic
    weights and inputs (tests/golden/nets_*.npz, written by scripts/pin_oracle_nets.py where `import diffusers` works).  While no
    such file exists -- diffusers is not installable in the build image -- the test SKIPS and the three forwards stay "parity
    unpinned" (DESIGN.md section 5); the day one is committed, this test is what turns that green."""
    import glob
    import os

    import pytest

    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "nets_*.npz")))
    if not files:
        pytest.skip("no tests/golden/nets_*.npz: scripts/pin_oracle_nets.py has not run where diffusers is installed (parity unpinned)")
    for path in files:
        z = np.load(path)
        tag = os.path.basename(path)[5:-4]
        ucfg, ccfg = (C.MINI_UNET, C.MINI_CONTROLNET) if tag == "mini" else (C.SD15_UNET, C.SD15_CONTROLNET)
        f32 = lambda w: {k: v.float() for k, v in w.items()}  # noqa: E731
        wu, wc, wv = (f32(W.synthesize(s, p)) for s, p in ((W.unet_spec(ucfg), "unet."), (W.controlnet_spec(ccfg), "cn."), (W.taesd_spec(C.TAESD), "vae.")))
        t = lambda k: torch.from_numpy(z[k])  # noqa: E731
        rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))  # noqa: E731
        with torch.no_grad():
            down, mid = nets.controlnet_forward(wc, ccfg, t("lat"), t("t"), t("text"), t("cond"), 1.0, True)
            for i, d in enumerate(down):
                assert rel(d, t(f"cn_down_{i}")) <= 1e-4, (tag, i)
            assert rel(mid, t("cn_mid")) <= 1e-4
            wemb = t("wemb") if z["wemb"].size else None
            eps = nets.unet_forward(wu, ucfg, t("lat"), t("t"), t("text"), wemb, [t(f"cn_down_{i}") for i in range(len(down))], t("cn_mid"))
            assert rel(eps, t("unet_eps")) <= 1e-4, tag
            assert rel(nets.taesd_encode(wv, t("img")), t("taesd_z")) <= 1e-4
            assert rel(nets.taesd_decode(wv, t("taesd_z")), t("taesd_x")) <= 1e-4
