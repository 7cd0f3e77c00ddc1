"""CLIP-L text encoder on the libvsd kernels (row A5 of SURVEY.md 8a).

Replaces `self.text_encoder(text_input_ids)` of the reference (/root/reference/diffusert/lcm/
lcm_controlnet.py:143-198), which diffusers re-runs on EVERY frame; here it runs once per prompt change and
the result is cached / broadcast (videosd_amd/dispatch.py).  12 pre-LN transformer layers, causal attention
(vsd_attention with causal=1), quick-GELU MLP, final LayerNorm; last_hidden_state [77, width] in fp16.
"""
import os
from typing import Dict, Optional

import torch

from . import lib as L
from .config import CLIPTextConfig
from .ops import Geom
from .packing import pack_linear, pack_linear_cat


class ClipTextEncoder:
    def __init__(self, ops, cfg: CLIPTextConfig, w: Dict[str, torch.Tensor], tokenizer_dir: Optional[str] = None):
        self.ops, self.cfg = ops, cfg
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

            self.tokenizer = CLIPTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))

    @property
    def has_tokenizer(self) -> bool:
        return self.tokenizer is not None

    def encode(self, text: str) -> torch.Tensor:
        ids = self.tokenizer(text, padding="max_length", max_length=self.cfg.max_len, truncation=True,
                             return_tensors="pt").input_ids[0]
        return self.encode_ids(ids)

    def encode_ids(self, ids: torch.Tensor) -> torch.Tensor:
        """ids: int64 [77] -> fp16 [77, width] on the device."""
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
        for ly in self.layers:
            ops.layernorm(x, s, c, ly["ln1"][0], ly["ln1"][1], cfg.eps, n)
            ops.conv(n, None, lin, ly["qkv"], qk, ldo=2 * c, out_t=vt, ldt=ldvt, t_col0=2 * c)
            ops.attention(qk, 2 * c, qk[:, c:], 2 * c, vt, ldvt, att, c, s, s, heads, d, d ** -0.5, causal=True)
            x2 = ops.empty(s, c)
            ops.conv(att, None, lin, ly["out"], x2, residual=x)
            ops.layernorm(x2, s, c, ly["ln2"][0], ly["ln2"][1], cfg.eps, n)
            ops.conv(n, None, lin, ly["fc1"], h, act=L.ACT_QUICKGELU)
            x = ops.empty(s, c)
            ops.conv(h, None, lin, ly["fc2"], x, residual=x2)
        out = ops.empty(s, c)
        ops.layernorm(x, s, c, self.final_ln[0], self.final_ln[1], cfg.eps, out)
        ops.synchronize()
        return out
