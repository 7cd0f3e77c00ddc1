"""CLIP text towers on the libvsd kernels (row A5 of SURVEY.md 8a; the SDXL pair of 8f row 3).

Replaces `self.text_encoder(text_input_ids)` of the reference (/root/reference/diffusert/lcm/
lcm_controlnet.py:143-198), which diffusers re-runs on EVERY frame; here it runs once per prompt change and
the result is cached / broadcast (videosd_amd/dispatch.py).  12 pre-LN transformer layers, causal attention
(vsd_attention with causal=1), quick-GELU MLP, final LayerNorm; last_hidden_state [77, width] in fp16.
"""
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import lib as L
from .config import CLIPTextConfig
from .ops import Geom
from .packing import pack_linear, pack_linear_cat
from .weights import Spec, clip_spec


@dataclass(frozen=True)
class TextTowerConfig(CLIPTextConfig):
    """The two text towers of an SDXL pipeline (BASELINE.json configs[3]; the reference itself runs one CLIP-L tower,
    lcm_controlnet.py:143-198).  `act`: the MLP's activation; `proj`: width of `text_projection` (0: the tower has none);
    `pad_id`: what the tower's tokenizer pads with (tokenizer_2 pads with "!" = id 0, CLIP-L's with end-of-text)."""
    act: str = "quick_gelu"
    proj: int = 0
    pad_id: int = 49407


SDXL_CLIP_L = TextTowerConfig()
SDXL_CLIP_G = TextTowerConfig(width=1280, heads=20, layers=32, mlp=5120, act="gelu", proj=1280, pad_id=0)  # 694 659 840 parameters
MINI_CLIP_G = TextTowerConfig(vocab=1000, width=128, heads=2, layers=3, mlp=512, act="gelu", proj=128, pad_id=0)


def text_tower_spec(cfg: CLIPTextConfig) -> Spec:
    """Tensor list of `CLIPTextModel` / `CLIPTextModelWithProjection` (transformers key names)."""
    spec = clip_spec(cfg)
    if getattr(cfg, "proj", 0):
        spec.append(("text_projection.weight", (cfg.proj, cfg.width), "w"))
    return spec


class ClipTextEncoder:
    def __init__(self, ops, cfg: CLIPTextConfig, w: Dict[str, torch.Tensor], tokenizer_dir: Optional[str] = None):
        self.ops, self.cfg = ops, cfg
        self.act = L.ACT_GELU if getattr(cfg, "act", "quick_gelu") == "gelu" else L.ACT_QUICKGELU
        self.proj = None
        if getattr(cfg, "proj", 0):
            self.proj = ops.to_device_pack(pack_linear(w["text_projection.weight"], None))
        t = "text_model"
        dev = ops.to_device
        self.tok_emb = dev(w[f"{t}.embeddings.token_embedding.weight"].half().contiguous())
        self.pos_emb = dev(w[f"{t}.embeddings.position_embedding.weight"].half().contiguous())
        self.layers = []
        for i in range(cfg.layers):
            p = f"{t}.encoder.layers.{i}"
            g = lambda n: (dev(w[f"{p}.{n}.weight"].half().contiguous()), dev(w[f"{p}.{n}.bias"].half().contiguous()))  # noqa: E731
            qkv = ops.to_device_pack(pack_linear_cat(
                [w[f"{p}.self_attn.{n}_proj.weight"] for n in ("q", "k", "v")],
                [w[f"{p}.self_attn.{n}_proj.bias"] for n in ("q", "k", "v")]))
            self.layers.append(dict(
                ln1=g("layer_norm1"), qkv=qkv,
                out=ops.to_device_pack(pack_linear(w[f"{p}.self_attn.out_proj.weight"], w[f"{p}.self_attn.out_proj.bias"])),
                ln2=g("layer_norm2"),
                fc1=ops.to_device_pack(pack_linear(w[f"{p}.mlp.fc1.weight"], w[f"{p}.mlp.fc1.bias"])),
                fc2=ops.to_device_pack(pack_linear(w[f"{p}.mlp.fc2.weight"], w[f"{p}.mlp.fc2.bias"]))))
        self.final_ln = (dev(w[f"{t}.final_layer_norm.weight"].half().contiguous()),
                         dev(w[f"{t}.final_layer_norm.bias"].half().contiguous()))
        self.tokenizer = None
        d = tokenizer_dir or os.environ.get("VSD_WEIGHTS")
        if d and os.path.exists(os.path.join(d, "vocab.json")) and os.path.exists(os.path.join(d, "merges.txt")):
            from transformers import CLIPTokenizer  # host-side BPE only; no model code from transformers

            pad = {} if getattr(cfg, "pad_id", 49407) != 0 else {"pad_token": "!"}  # (tokenizer_2 of SDXL: id 0)
            self.tokenizer = CLIPTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"), **pad)

    @property
    def has_tokenizer(self) -> bool:
        return self.tokenizer is not None

    def tokenize(self, text: str) -> torch.Tensor:
        return self.tokenizer(text, padding="max_length", max_length=self.cfg.max_len, truncation=True,
                              return_tensors="pt").input_ids[0]

    def encode(self, text: str) -> torch.Tensor:
        return self.encode_ids(self.tokenize(text))

    def encode_ids(self, ids: torch.Tensor) -> torch.Tensor:
        """ids: int64 [77] -> fp16 [77, width] on the device: `text_encoder(ids)[0]` (lcm_controlnet.py:175)."""
        return self.hidden_states(ids)[1]

    def text_embeds(self, ids: torch.Tensor, last_normed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`CLIPTextModelWithProjection(ids).text_embeds` [proj]: the first end-of-text row (= the first maximum of the ids) of
        the normed last state through `text_projection`.  (All 77 rows go through the GEMM -- one 64-row tile more than the one
        row needed, once per prompt -- and the row is picked afterwards.)"""
        if self.proj is None:
            raise RuntimeError("this text tower has no text_projection")
        ops, s = self.ops, ids.numel()
        if last_normed is None:
            last_normed = self.hidden_states(ids)[1]
        rows = ops.empty(s, self.proj.n)
        ops.conv(last_normed, None, Geom.linear(s), self.proj, rows)
        ops.synchronize()
        return rows[int(ids.reshape(-1).argmax())].clone()

    def hidden_states(self, ids: torch.Tensor):
        """ids: int64 [77] -> (hidden_states[-2] [77, width]: the state BEFORE the last layer, what an SDXL pipeline
        conditions on; final_layer_norm(hidden_states[-1]) [77, width]: `text_encoder(ids)[0]`), fp16 on the device."""
        ops, cfg = self.ops, self.cfg
        s, c, heads = ids.numel(), cfg.width, cfg.heads
        d = c // heads
        ids = ops.to_device(ids.reshape(-1).long().contiguous())
        x = ops.empty(s, c)
        ops.embed_tokens(ids, self.tok_emb, self.pos_emb, x)  # token gather + position add (vsd_embed_tokens)
        lin = Geom.linear(s)
        ldvt = (s + 63) // 64 * 64
        n = ops.empty(s, c)
        qk = ops.empty(s, 2 * c)
        vt = ops.zeros(c, ldvt)
        att = ops.empty(s, c)
        h = ops.empty(s, cfg.mlp)
        penultimate = None
        for li, ly in enumerate(self.layers):
            if li == len(self.layers) - 1:
                penultimate = x
            ops.layernorm(x, s, c, ly["ln1"][0], ly["ln1"][1], cfg.eps, n)
            ops.conv(n, None, lin, ly["qkv"], qk, ldo=2 * c, out_t=vt, ldt=ldvt, t_col0=2 * c)
            ops.attention(qk, 2 * c, qk[:, c:], 2 * c, vt, ldvt, att, c, s, s, heads, d, d ** -0.5, causal=True)
            x2 = ops.empty(s, c)
            ops.conv(att, None, lin, ly["out"], x2, residual=x)
            ops.layernorm(x2, s, c, ly["ln2"][0], ly["ln2"][1], cfg.eps, n)
            ops.conv(n, None, lin, ly["fc1"], h, act=self.act)
            x = ops.empty(s, c)
            ops.conv(h, None, lin, ly["fc2"], x, residual=x2)
        out = ops.empty(s, c)
        ops.layernorm(x, s, c, self.final_ln[0], self.final_ln[1], cfg.eps, out)
        ops.synchronize()
        return penultimate, out


class SdxlTextEncoders:
    """What `StableDiffusionXLPipeline.encode_prompt` computes (diffusers 0.23-0.25, the generation SURVEY.md brackets the
    reference to): prompt_embeds = cat(tower1.hidden_states[-2], tower2.hidden_states[-2]) [77, 2048] and
    pooled_prompt_embeds = tower2.text_embeds [1280] -- both towers on the libvsd kernels, once per prompt."""

    def __init__(self, tower1: ClipTextEncoder, tower2: ClipTextEncoder):
        self.t1, self.t2 = tower1, tower2
        if tower2.proj is None:
            raise ValueError("the second SDXL text tower needs text_projection.weight")

    @property
    def has_tokenizer(self) -> bool:
        return self.t1.has_tokenizer and self.t2.has_tokenizer

    def encode_ids(self, ids1: torch.Tensor, ids2: torch.Tensor):
        h1, _ = self.t1.hidden_states(ids1)
        h2, last2 = self.t2.hidden_states(ids2)
        return torch.cat([h1, h2], dim=1).contiguous(), self.t2.text_embeds(ids2, last2)

    def encode(self, text: str):
        return self.encode_ids(self.t1.tokenize(text), self.t2.tokenize(text))
