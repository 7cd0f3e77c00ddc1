"""fp32 CPU restatement of the neural modules the reference calls (TEST INFRASTRUCTURE, see oracle/__init__).

The reference obtains these from `diffusers` / `transformers` (call sites
/root/reference/diffusert/lcm/lcm_controlnet.py:175 CLIP, :299 TAESD encode, :558 ControlNet,
:568 UNet, :594 TAESD decode).  Architecture: SURVEY.md Appendix A.  All functions are purely
functional over a dict of tensors keyed by the diffusers parameter names (videosd_amd/weights.py),
NCHW fp32.
"""
import math

import torch
import torch.nn.functional as F


def _f(w, name):
    return w[name].float()


# fp16-STORAGE emulation (off by default: the oracle is fp32).  When set, the output of every conv / linear / norm layer and the
# attention probabilities are rounded to fp16 and back -- what ANY fp16 implementation of the path does to its activations (the
# reference's own fp16 diffusers pipeline included) while accumulating in fp32.  Used to separate "the HIP path differs from the
# algorithm" from "fp16 storage moves this network's output by that much" on weight sets where the latter is large (the
# range-stress set: tests/test_pipeline_gpu.py::test_baseline_config2_on_range_stress_weights_matches_oracle).
EMULATE_FP16 = False


def _r(x):
    return x.half().float() if EMULATE_FP16 else x


def conv(w, name, x, stride=1, padding=1):
    b = w.get(name + ".bias")
    return _r(F.conv2d(x, _f(w, name + ".weight"), None if b is None else b.float(), stride=stride, padding=padding))


def linear(w, name, x):
    b = w.get(name + ".bias")
    return _r(F.linear(x, _f(w, name + ".weight"), None if b is None else b.float()))


def group_norm(w, name, x, groups, eps):
    return _r(F.group_norm(x, groups, _f(w, name + ".weight"), _f(w, name + ".bias"), eps))


def layer_norm(w, name, x, eps=1e-5):
    return _r(F.layer_norm(x, (x.shape[-1],), _f(w, name + ".weight"), _f(w, name + ".bias"), eps))


def timestep_sinusoid(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def time_embedding(w, cfg, t, w_emb=None, added=None):
    t_emb = timestep_sinusoid(t, cfg.block_out_channels[0])
    if w_emb is not None:
        t_emb = t_emb + linear(w, "time_embedding.cond_proj", w_emb)
    h = linear(w, "time_embedding.linear_1", t_emb)
    emb = linear(w, "time_embedding.linear_2", F.silu(h))
    if cfg.add_time_dim:
        # SDXL addition_embed_type="text_time" (UNet2DConditionModel.forward): aug_emb = add_embedding(cat[pooled text
        # embeds, flattened sinusoids of the 6 time ids]); emb = emb + aug_emb
        pooled, time_ids = added
        tid = timestep_sinusoid(time_ids.flatten(), cfg.add_time_dim).reshape(time_ids.shape[0], -1)
        a = torch.cat([pooled.float(), tid], dim=-1)
        a = linear(w, "add_embedding.linear_2", F.silu(linear(w, "add_embedding.linear_1", a)))
        emb = emb + a
    return emb


def resnet(w, p, cfg, x, temb):
    h = F.silu(group_norm(w, p + ".norm1", x, cfg.groups, 1e-5))
    h = conv(w, p + ".conv1", h)
    h = h + linear(w, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(group_norm(w, p + ".norm2", h, cfg.groups, 1e-5))
    h = conv(w, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in w:
        x = conv(w, p + ".conv_shortcut", x, padding=0)
    return x + h


def attention(w, p, x, ctx, heads):
    b, s, c = x.shape
    q = linear(w, p + ".to_q", x)
    k = linear(w, p + ".to_k", ctx)
    v = linear(w, p + ".to_v", ctx)
    d = c // heads

    def split(t):
        return t.view(b, -1, heads, d).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    att = _r(torch.softmax(q @ k.transpose(-1, -2) * (d ** -0.5), dim=-1))
    o = (att @ v).transpose(1, 2).reshape(b, s, c)
    return linear(w, p + ".to_out.0", o)


class RefState:
    """The state of the "reference-only" mode (dead code at the reference's v2, kept behind `ref` / `style_fidelity`:
    /root/reference/diffusert/lcm/lcm_reference_pipeline.py:488-853).  Per denoising step the UNet runs twice: a WRITE
    pass over the noised reference latents that banks (a) every BasicTransformerBlock's norm1 output (:527-528) and
    (b) the per-channel spatial mean / variance of the block outputs of the gated blocks (:587-598, :621-625, ...), then
    the READ pass over the frame's latents, whose self-attentions see [x ; bank] as keys / values (:535-546) and whose
    gated block outputs are re-normalised to the banked statistics (AdaIN, :593-603).  `style_fidelity` blends two
    identical tensors there (`x_c = x_uc.clone()`, :541, :601) and so has no effect."""

    def __init__(self, n_down, n_up, attention_auto_machine_weight=1.0, gn_auto_machine_weight=1.0):
        self.mode = "write"
        self.attn_bank = {}   # transformer-block name -> norm1 output of the write pass
        self.stat_bank = {}   # (block, layer) -> (mean, var) of the write pass
        self.attn_w = attention_auto_machine_weight
        self.gn_w = gn_auto_machine_weight
        self.n_down, self.n_up = n_down, n_up
        self.attn_rank = {}   # transformer-block name -> attn_weight (:806-813), filled by `rank_attention`

    def gn_gate(self, kind, i):
        """module.gn_weight (:816-846): mid 0; down w -> 2 (1 - w / n_down); up w -> 2 w / n_up; banked iff
        gn_auto_machine_weight >= gn_weight."""
        gw = 0.0 if kind == "mid" else (2.0 * (1.0 - i / self.n_down) if kind == "down" else 2.0 * i / self.n_up)
        return self.gn_w >= gw

    def adain(self, key, x, eps=1e-6):
        if self.mode == "write":
            var, mean = torch.var_mean(x, dim=(2, 3), keepdim=True, correction=0)
            self.stat_bank[key] = (mean, var)
            return x
        if key not in self.stat_bank:
            return x
        mean_acc, var_acc = self.stat_bank.pop(key)
        var, mean = torch.var_mean(x, dim=(2, 3), keepdim=True, correction=0)
        std = torch.maximum(var, torch.zeros_like(var) + eps) ** 0.5
        std_acc = torch.maximum(var_acc, torch.zeros_like(var_acc) + eps) ** 0.5
        return ((x - mean) / std) * std_acc + mean_acc


def transformer_block_names(cfg):
    """Every BasicTransformerBlock of the UNet with its channel width, in module order."""
    out = []
    ch = cfg.block_out_channels
    for i in range(len(ch)):
        if cfg.down_attn[i]:
            for j in range(cfg.layers_per_block):
                out += [(f"down_blocks.{i}.attentions.{j}.transformer_blocks.{k}", ch[i]) for k in range(cfg.transformer_depth[i])]
    out += [(f"mid_block.attentions.0.transformer_blocks.{k}", ch[-1]) for k in range(cfg.mid_depth)]
    rev = list(reversed(ch))
    for i in range(len(ch)):
        if cfg.up_attn[i]:
            for j in range(cfg.layers_per_block + 1):
                out += [(f"up_blocks.{i}.attentions.{j}.transformer_blocks.{k}", rev[i]) for k in range(cfg.up_depth[i])]
    return out


def rank_attention(cfg, ref: RefState):
    """attn_weight = rank / count after a stable sort by descending width (:806-813); a block reads the bank iff
    attention_auto_machine_weight > attn_weight (:529)."""
    names = transformer_block_names(cfg)
    order = sorted(range(len(names)), key=lambda i: -names[i][1])
    for rank, i in enumerate(order):
        ref.attn_rank[names[i][0]] = rank / float(len(names))


def transformer(w, p, cfg, x, text, depth=1, ref=None):
    """Transformer2DModel: `depth` BasicTransformerBlocks between proj_in / proj_out (1x1 conv for SD1.5, Linear on
    the token matrix when use_linear_projection, SDXL)."""
    b, c, hh, ww = x.shape
    heads = cfg.heads_for(c)
    res = x
    h = group_norm(w, p + ".norm", x, cfg.groups, 1e-6)
    if cfg.linear_proj:
        h = linear(w, p + ".proj_in", h.permute(0, 2, 3, 1).reshape(b, hh * ww, c))
    else:
        h = conv(w, p + ".proj_in", h, padding=0)
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    for k in range(depth):
        t = f"{p}.transformer_blocks.{k}"
        n = layer_norm(w, t + ".norm1", h)
        kv = n
        if ref is not None:
            if ref.mode == "write":
                ref.attn_bank[t] = n.clone()
            elif ref.attn_w > ref.attn_rank.get(t, 0.0) and t in ref.attn_bank:
                kv = torch.cat([n, ref.attn_bank.pop(t)], dim=1)
        h = h + attention(w, t + ".attn1", n, kv, heads)
        n = layer_norm(w, t + ".norm2", h)
        h = h + attention(w, t + ".attn2", n, text, heads)
        n = layer_norm(w, t + ".norm3", h)
        g = linear(w, t + ".ff.net.0.proj", n)
        hid, gate = g.chunk(2, dim=-1)
        h = h + linear(w, t + ".ff.net.2", hid * F.gelu(gate))
    if cfg.linear_proj:
        h = linear(w, p + ".proj_out", h).reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    else:
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        h = conv(w, p + ".proj_out", h, padding=0)
    return h + res


def _down_and_mid(w, cfg, h, temb, text, ref=None):
    skips = [h]
    ch = cfg.block_out_channels
    for i in range(len(ch)):
        for j in range(cfg.layers_per_block):
            h = resnet(w, f"down_blocks.{i}.resnets.{j}", cfg, h, temb)
            if cfg.down_attn[i]:
                h = transformer(w, f"down_blocks.{i}.attentions.{j}", cfg, h, text, cfg.transformer_depth[i], ref)
            if ref is not None and ref.gn_gate("down", i):
                h = ref.adain(("down", i, j), h)
            skips.append(h)
        if i < len(ch) - 1:
            h = conv(w, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)
    h = resnet(w, "mid_block.resnets.0", cfg, h, temb)
    h = transformer(w, "mid_block.attentions.0", cfg, h, text, cfg.mid_depth, ref)
    h = resnet(w, "mid_block.resnets.1", cfg, h, temb)
    if ref is not None and ref.gn_gate("mid", 0):
        h = ref.adain(("mid", 0, 0), h)
    return h, skips


def unet_forward(w, cfg, sample, t, text, w_emb=None, down_res=None, mid_res=None, added=None, ref=None):
    """UNet2DConditionModel.forward as used at lcm_controlnet.py:568-577.  t: int64 [B].
    added = (pooled text embeds [B, add_pooled_dim], time ids [B, 6]) for the SDXL configuration.
    ref: a RefState in "write" or "read" mode (reference-only extension, see RefState)."""
    temb = time_embedding(w, cfg, t, w_emb, added)
    h = conv(w, "conv_in", sample)
    h, skips = _down_and_mid(w, cfg, h, temb, text, ref)
    if down_res is not None:
        skips = [s + r for s, r in zip(skips, down_res)]
        h = h + mid_res
    # diffusers: any latent dim not divisible by 2**(num_upsamplers) -> upsample to the skip's size
    n_up = len(cfg.block_out_channels) - 1
    fwd_size = any(d % (2 ** n_up) != 0 for d in sample.shape[-2:])
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(w, f"up_blocks.{i}.resnets.{j}", cfg, h, temb)
            if cfg.up_attn[i]:
                h = transformer(w, f"up_blocks.{i}.attentions.{j}", cfg, h, text, cfg.up_depth[i], ref)
            if ref is not None and ref.gn_gate("up", i):
                h = ref.adain(("up", i, j), h)
        if i < nb - 1:
            if fwd_size:
                h = F.interpolate(h, size=skips[-1].shape[2:], mode="nearest")
            else:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv(w, f"up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(group_norm(w, "conv_norm_out", h, cfg.groups, 1e-5))
    return conv(w, "conv_out", h)


def controlnet_cond_embedding(w, ccfg, cond):
    h = F.silu(conv(w, "controlnet_cond_embedding.conv_in", cond))
    nblk = 2 * (len(ccfg.cond_channels) - 1)
    for k in range(nblk):
        h = F.silu(conv(w, f"controlnet_cond_embedding.blocks.{k}", h, stride=2 if k % 2 == 1 else 1))
    return conv(w, "controlnet_cond_embedding.conv_out", h)


def controlnet_forward(w, ccfg, sample, t, text, cond, conditioning_scale=1.0, guess_mode=True):
    """ControlNetModel.forward as used at lcm_controlnet.py:558-566 (guess_mode always True there)."""
    cfg = ccfg.unet
    temb = time_embedding(w, cfg, t, None)
    h = conv(w, "conv_in", sample) + controlnet_cond_embedding(w, ccfg, cond)
    h, skips = _down_and_mid(w, cfg, h, temb, text)
    down = [conv(w, f"controlnet_down_blocks.{i}", s, padding=0) for i, s in enumerate(skips)]
    mid = conv(w, "controlnet_mid_block", h, padding=0)
    if guess_mode:
        scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
        down = [d * s for d, s in zip(down, scales)]
        mid = mid * scales[-1]
    else:
        down = [d * conditioning_scale for d in down]
        mid = mid * conditioning_scale
    return down, mid


def _taesd_block(w, p, x):
    h = F.relu(conv(w, p + ".conv.0", x))
    h = F.relu(conv(w, p + ".conv.2", h))
    h = conv(w, p + ".conv.4", h)
    return F.relu(h + x)


def taesd_encode(w, x):
    """AutoencoderTiny.encode(x).latents; x in [-1,1] NCHW."""
    e = "encoder.layers"
    h = conv(w, f"{e}.0", (x + 1) / 2)
    h = _taesd_block(w, f"{e}.1", h)
    n = 2
    for _ in range(3):
        h = conv(w, f"{e}.{n}", h, stride=2)
        n += 1
        for _ in range(3):
            h = _taesd_block(w, f"{e}.{n}", h)
            n += 1
    return conv(w, f"{e}.{n}", h)


def taesd_decode(w, z):
    """AutoencoderTiny.decode(z).sample; returns image in ~[-1,1]."""
    d = "decoder.layers"
    h = torch.tanh(z / 3) * 3
    h = F.relu(conv(w, f"{d}.0", h))
    n = 2
    for nb in (3, 3, 3):
        for _ in range(nb):
            h = _taesd_block(w, f"{d}.{n}", h)
            n += 1
        h = F.interpolate(h, scale_factor=2.0, mode="nearest")
        n += 1
        h = conv(w, f"{d}.{n}", h)
        n += 1
    h = _taesd_block(w, f"{d}.{n}", h)
    n += 1
    h = conv(w, f"{d}.{n}", h)
    return h * 2 - 1


def clip_text_forward(w, cfg, ids):
    """CLIPTextModel(ids)[0] (last_hidden_state): causal mask, quick-GELU, final LayerNorm."""
    t = "text_model"
    b, s = ids.shape
    x = _f(w, f"{t}.embeddings.token_embedding.weight")[ids] + _f(w, f"{t}.embeddings.position_embedding.weight")[:s]
    mask = torch.full((s, s), float("-inf")).triu(1)
    d = cfg.width // cfg.heads
    for i in range(cfg.layers):
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
        h = h * torch.sigmoid(1.702 * h)
        x = x + linear(w, p + ".mlp.fc2", h)
    return layer_norm(w, f"{t}.final_layer_norm", x, cfg.eps)
