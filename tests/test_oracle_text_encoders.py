"""The SDXL prompt side: the oracle restatement of the two text towers pinned against `transformers`
(where the reference's arithmetic for this lives: requirements.txt:3), the architecture anchored by its published parameter
count, and the host logic of videosd_amd/clip.py over the op emulator."""
import torch

from oracle import text_encoders as T
from videosd_amd import clip as K
from videosd_amd import weights as W


def _ids(cfg, eos_at, seed):
    """begin-of-text, random tokens, end-of-text (the largest id) at `eos_at`, then the tower's padding"""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, cfg.vocab - 1, (cfg.max_len,), generator=g)
    ids[eos_at] = cfg.vocab - 1
    ids[eos_at + 1:] = cfg.pad_id if cfg.pad_id < cfg.vocab else cfg.vocab - 1
    return ids


def _hf(cfg, with_projection):
    import transformers

    hf_cfg = transformers.CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.width, intermediate_size=cfg.mlp,
                                         num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                                         max_position_embeddings=cfg.max_len, hidden_act=cfg.act, layer_norm_eps=cfg.eps,
                                         projection_dim=cfg.proj or 512, bos_token_id=0, eos_token_id=cfg.vocab - 1)
    cls = transformers.CLIPTextModelWithProjection if with_projection else transformers.CLIPTextModel
    return cls(hf_cfg).eval().float()


def _load(m, w):
    keys = set(m.state_dict().keys())  # (newer transformers drop the "text_model." level from the module tree; checkpoints keep it)
    if not any(k.startswith("text_model.") for k in keys):
        w = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in w.items()}
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)


def test_bigg_tower_has_the_published_parameter_count():
    assert W.count_params(K.text_tower_spec(K.SDXL_CLIP_G)) == 694_659_840   # CLIPTextModelWithProjection of SDXL-base
    assert W.count_params(K.text_tower_spec(K.SDXL_CLIP_L)) == 123_060_480


def test_penultimate_states_and_text_embeds_match_transformers():
    c1 = K.TextTowerConfig(vocab=1000, width=128, heads=2, layers=2, mlp=512)
    c2 = K.MINI_CLIP_G
    w1 = W.synthesize(K.text_tower_spec(c1), "t1.", dtype=torch.float32)
    w2 = W.synthesize(K.text_tower_spec(c2), "t2.", dtype=torch.float32)
    m1, m2 = _hf(c1, False), _hf(c2, True)
    _load(m1, w1)
    _load(m2, w2)
    for eos_at, seed in ((5, 1), (40, 2), (76, 3)):
        i1, i2 = _ids(c1, eos_at, seed)[None], _ids(c2, eos_at, seed + 10)[None]
        with torch.no_grad():
            r1 = m1(i1, output_hidden_states=True)
            r2 = m2(i2, output_hidden_states=True)
        emb, pooled = T.sdxl_prompt_embeds(w1, c1, i1, w2, c2, i2)
        want = torch.cat([r1.hidden_states[-2], r2.hidden_states[-2]], dim=-1)
        assert torch.allclose(emb, want, atol=3e-5, rtol=1e-4), float((emb - want).abs().max())
        assert torch.allclose(pooled, r2.text_embeds, atol=3e-5, rtol=1e-4), float((pooled - r2.text_embeds).abs().max())
        # and the single-tower call of the reference (lcm_controlnet.py:175) is the second return value
        assert torch.allclose(T.clip_text_hidden(w1, c1, i1)[1], r1[0], atol=3e-5, rtol=1e-4)


def test_sdxl_text_encoders_host_logic_matches_the_oracle():
    from fake_ops import FakeOps

    c1 = K.TextTowerConfig(vocab=1000, width=128, heads=2, layers=2, mlp=512)
    c2 = K.MINI_CLIP_G
    w1, w2 = W.synthesize(K.text_tower_spec(c1), "t1."), W.synthesize(K.text_tower_spec(c2), "t2.")
    ops = FakeOps()
    enc = K.SdxlTextEncoders(K.ClipTextEncoder(ops, c1, w1), K.ClipTextEncoder(ops, c2, w2))
    assert not enc.has_tokenizer
    i1, i2 = _ids(c1, 9, 4), _ids(c2, 9, 5)
    emb, pooled = enc.encode_ids(i1, i2)
    want_e, want_p = T.sdxl_prompt_embeds(w1, c1, i1[None], w2, c2, i2[None])
    assert emb.shape == (77, 256) and pooled.shape == (128,)
    assert float((emb.float() - want_e[0]).norm() / want_e.norm()) < 5e-3
    assert float((pooled.float() - want_p[0]).norm() / want_p.norm()) < 5e-3
    # a tower without text_projection cannot be the second one
    try:
        K.SdxlTextEncoders(enc.t2, enc.t1)
        raise AssertionError("accepted a second tower without text_projection")
    except ValueError:
        pass
