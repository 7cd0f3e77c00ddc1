"""TEST INFRASTRUCTURE -- CPU restatement of the prompt side of BASELINE.json configs[3] (SDXL-base).

The reference (/root/reference/diffusert/lcm/lcm_controlnet.py:143-198) encodes its prompt with ONE CLIP text tower and takes
`text_encoder(ids)[0]`; an SDXL pipeline of the same diffusers generation (StableDiffusionXLPipeline.encode_prompt, diffusers
0.23-0.25, the bracket SURVEY.md names) runs TWO towers and takes
    prompt_embeds        = cat(tower1.hidden_states[-2], tower2.hidden_states[-2], dim=-1)      [77, 768 + 1280]
    pooled_prompt_embeds = tower2.text_embeds = text_projection(final_layer_norm(last)[eos])    [1280]
tower 1 = CLIP-L (quick-GELU), tower 2 = OpenCLIP bigG as `CLIPTextModelWithProjection` (erf GELU, 32 layers, 1280 wide, 20 heads,
pad token id 0).  The arithmetic lives in `transformers` (unvendored, unpinned: requirements.txt:3), which IS installed here:
tests/test_oracle_text_encoders.py pins every function below against `transformers.CLIPTextModel` /
`CLIPTextModelWithProjection` built from the same tensors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
import torch
import torch.nn.functional as F

from .nets import _f, layer_norm, linear


def clip_text_hidden(w, cfg, ids):
    """ids [B, S] -> (hidden_states[-2], final_layer_norm(hidden_states[-1])): the tower of nets.clip_text_forward with the
    activation of `cfg.act` ("quick_gelu" | "gelu") and the state BEFORE the last layer kept."""
    t = "text_model"
    b, s = ids.shape
    x = _f(w, f"{t}.embeddings.token_embedding.weight")[ids] + _f(w, f"{t}.embeddings.position_embedding.weight")[:s]
    mask = torch.full((s, s), float("-inf")).triu(1)
    d = cfg.width // cfg.heads
    act = getattr(cfg, "act", "quick_gelu")
    penultimate = None
    for i in range(cfg.layers):
        if i == cfg.layers - 1:
            penultimate = x
        p = f"{t}.encoder.layers.{i}"
        n = layer_norm(w, p + ".layer_norm1", x, cfg.eps)
        q = linear(w, p + ".self_attn.q_proj", n).view(b, s, cfg.heads, d).transpose(1, 2)
        k = linear(w, p + ".self_attn.k_proj", n).view(b, s, cfg.heads, d).transpose(1, 2)
        v = linear(w, p + ".self_attn.v_proj", n).view(b, s, cfg.heads, d).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) * (d ** -0.5) + mask, dim=-1)
        o = (att @ v).transpose(1, 2).reshape(b, s, cfg.width)
        x = x + linear(w, p + ".self_attn.out_proj", o)
        n = layer_norm(w, p + ".layer_norm2", x, cfg.eps)
        h = linear(w, p + ".mlp.fc1", n)
        h = F.gelu(h) if act == "gelu" else h * torch.sigmoid(1.702 * h)
        x = x + linear(w, p + ".mlp.fc2", h)
    return penultimate, layer_norm(w, f"{t}.final_layer_norm", x, cfg.eps)


def eos_index(ids):
    """The pooled row of CLIPTextTransformer: the first end-of-text token = the first maximum of the ids (49407 is the largest id
    of both SDXL vocabularies)."""
    return ids.argmax(dim=-1)


def clip_text_embeds(w, cfg, ids):
    """CLIPTextModelWithProjection(ids).text_embeds: the end-of-text row of the normed last state through `text_projection`
    (no bias)."""
    _, last = clip_text_hidden(w, cfg, ids)
    pooled = last[torch.arange(ids.shape[0]), eos_index(ids)]
    return pooled @ _f(w, "text_projection.weight").t()


def sdxl_prompt_embeds(w1, cfg1, ids1, w2, cfg2, ids2):
    """-> (prompt_embeds [B, S, width1 + width2], pooled_prompt_embeds [B, proj2])"""
    h1, _ = clip_text_hidden(w1, cfg1, ids1)
    h2, _ = clip_text_hidden(w2, cfg2, ids2)
    return torch.cat([h1, h2], dim=-1), clip_text_embeds(w2, cfg2, ids2)
