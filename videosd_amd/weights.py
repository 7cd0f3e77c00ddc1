"""Weight containers keyed by diffusers / transformers parameter names.

Real checkpoints (safetensors with these key names) load unchanged through `load_safetensors`.
When none exist (no network in the build image) `synthesize` fills every tensor from a generator
seeded by the tensor's name, so the CPU oracle and the HIP engine always see identical values.

Replaces: the `from_pretrained` calls of /root/reference/diffusert/videopipeline.py:51-69.
"""
import zlib
from typing import Dict, List, Tuple

import torch

from .config import CLIPTextConfig, ControlNetConfig, TAESDConfig, UNetConfig

Spec = List[Tuple[str, Tuple[int, ...], str]]  # (name, shape, kind)


def _conv(spec: Spec, name: str, cout: int, cin: int, k: int, bias: bool = True):
    spec.append((name + ".weight", (cout, cin, k, k), "w"))
    if bias:
        spec.append((name + ".bias", (cout,), "b"))


def _lin(spec: Spec, name: str, cout: int, cin: int, bias: bool = True):
    spec.append((name + ".weight", (cout, cin), "w"))
    if bias:
        spec.append((name + ".bias", (cout,), "b"))


def _norm(spec: Spec, name: str, c: int):
    spec.append((name + ".weight", (c,), "g"))
    spec.append((name + ".bias", (c,), "b"))


def _resnet(spec: Spec, p: str, cin: int, cout: int, temb: int):
    _norm(spec, p + ".norm1", cin)
    _conv(spec, p + ".conv1", cout, cin, 3)
    _lin(spec, p + ".time_emb_proj", cout, temb)
    _norm(spec, p + ".norm2", cout)
    _conv(spec, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(spec, p + ".conv_shortcut", cout, cin, 1)


def _transformer(spec: Spec, p: str, c: int, cross: int, depth: int = 1, linear_proj: bool = False):
    _norm(spec, p + ".norm", c)
    if linear_proj:
        _lin(spec, p + ".proj_in", c, c)
    else:
        _conv(spec, p + ".proj_in", c, c, 1)
    for k in range(depth):
        b = f"{p}.transformer_blocks.{k}"
        _norm(spec, b + ".norm1", c)
        for n in ("to_q", "to_k", "to_v"):
            _lin(spec, f"{b}.attn1.{n}", c, c, bias=False)
        _lin(spec, b + ".attn1.to_out.0", c, c)
        _norm(spec, b + ".norm2", c)
        _lin(spec, b + ".attn2.to_q", c, c, bias=False)
        _lin(spec, b + ".attn2.to_k", c, cross, bias=False)
        _lin(spec, b + ".attn2.to_v", c, cross, bias=False)
        _lin(spec, b + ".attn2.to_out.0", c, c)
        _norm(spec, b + ".norm3", c)
        _lin(spec, b + ".ff.net.0.proj", 8 * c, c)
        _lin(spec, b + ".ff.net.2", c, 4 * c)
    if linear_proj:
        _lin(spec, p + ".proj_out", c, c)
    else:
        _conv(spec, p + ".proj_out", c, c, 1)


def _encoder_half(spec: Spec, cfg: UNetConfig):
    """conv_in, time embedding, down blocks, mid block (shared by UNet and ControlNet)."""
    ch = cfg.block_out_channels
    _conv(spec, "conv_in", ch[0], cfg.in_channels, 3)
    _lin(spec, "time_embedding.linear_1", cfg.temb_dim, ch[0])
    _lin(spec, "time_embedding.linear_2", cfg.temb_dim, cfg.temb_dim)
    if cfg.cond_proj_dim:
        _lin(spec, "time_embedding.cond_proj", ch[0], cfg.cond_proj_dim, bias=False)
    if cfg.add_time_dim:  # SDXL text_time conditioning: TimestepEmbedding(pooled + 6 sinusoids -> temb_dim)
        _lin(spec, "add_embedding.linear_1", cfg.temb_dim, cfg.add_in_dim)
        _lin(spec, "add_embedding.linear_2", cfg.temb_dim, cfg.temb_dim)
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _resnet(spec, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, cfg.temb_dim)
            if cfg.down_attn[i]:
                _transformer(spec, f"down_blocks.{i}.attentions.{j}", cout, cfg.cross_dim, cfg.transformer_depth[i],
                             cfg.linear_proj)
        if i < len(ch) - 1:
            _conv(spec, f"down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
        cin = cout
    _resnet(spec, "mid_block.resnets.0", ch[-1], ch[-1], cfg.temb_dim)
    _transformer(spec, "mid_block.attentions.0", ch[-1], cfg.cross_dim, cfg.mid_depth, cfg.linear_proj)
    _resnet(spec, "mid_block.resnets.1", ch[-1], ch[-1], cfg.temb_dim)


def skip_channels(cfg: UNetConfig) -> List[int]:
    """Channel count of each skip tensor (12 for SD1.5, 9 for SDXL) in push order."""
    ch = cfg.block_out_channels
    out = [ch[0]]
    for i, c in enumerate(ch):
        out += [c] * cfg.layers_per_block
        if i < len(ch) - 1:
            out.append(c)
    return out


def unet_spec(cfg: UNetConfig) -> Spec:
    spec: Spec = []
    _encoder_half(spec, cfg)
    ch = cfg.block_out_channels
    skips = skip_channels(cfg)
    rev = list(reversed(ch))
    prev = ch[-1]
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            sc = skips.pop()
            cin = (prev if j == 0 else cout) + sc
            _resnet(spec, f"up_blocks.{i}.resnets.{j}", cin, cout, cfg.temb_dim)
            if cfg.up_attn[i]:
                _transformer(spec, f"up_blocks.{i}.attentions.{j}", cout, cfg.cross_dim, cfg.up_depth[i], cfg.linear_proj)
        if i < len(rev) - 1:
            _conv(spec, f"up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
        prev = cout
    _norm(spec, "conv_norm_out", ch[0])
    _conv(spec, "conv_out", cfg.out_channels, ch[0], 3)
    return spec


def controlnet_spec(cfg: ControlNetConfig) -> Spec:
    spec: Spec = []
    _encoder_half(spec, cfg.unet)
    cc = cfg.cond_channels
    _conv(spec, "controlnet_cond_embedding.conv_in", cc[0], cfg.cond_in, 3)
    k = 0
    for i in range(len(cc) - 1):
        _conv(spec, f"controlnet_cond_embedding.blocks.{k}", cc[i], cc[i], 3)
        _conv(spec, f"controlnet_cond_embedding.blocks.{k + 1}", cc[i + 1], cc[i], 3)
        k += 2
    _conv(spec, "controlnet_cond_embedding.conv_out", cfg.unet.block_out_channels[0], cc[-1], 3)
    for i, c in enumerate(skip_channels(cfg.unet)):
        _conv(spec, f"controlnet_down_blocks.{i}", c, c, 1)
    c = cfg.unet.block_out_channels[-1]
    _conv(spec, "controlnet_mid_block", c, c, 1)
    return spec


def _taesd_block(spec: Spec, p: str, c: int):
    for k in (0, 2, 4):
        _conv(spec, f"{p}.conv.{k}", c, c, 3)


def taesd_spec(cfg: TAESDConfig) -> Spec:
    """encoder.layers.N / decoder.layers.N numbering of diffusers' EncoderTiny / DecoderTiny."""
    spec: Spec = []
    c = cfg.channels
    e = "encoder.layers"
    _conv(spec, f"{e}.0", c, cfg.image_channels, 3)
    _taesd_block(spec, f"{e}.1", c)
    n = 2
    for _ in range(3):
        _conv(spec, f"{e}.{n}", c, c, 3, bias=False)
        n += 1
        for _ in range(3):
            _taesd_block(spec, f"{e}.{n}", c)
            n += 1
    _conv(spec, f"{e}.{n}", cfg.latent_channels, c, 3)
    d = "decoder.layers"
    _conv(spec, f"{d}.0", c, cfg.latent_channels, 3)
    n = 2  # layer 1 is the ReLU
    for nb in (3, 3, 3):
        for _ in range(nb):
            _taesd_block(spec, f"{d}.{n}", c)
            n += 1
        n += 1  # nn.Upsample
        _conv(spec, f"{d}.{n}", c, c, 3, bias=False)
        n += 1
    _taesd_block(spec, f"{d}.{n}", c)
    n += 1
    _conv(spec, f"{d}.{n}", cfg.image_channels, c, 3)
    return spec


def clip_spec(cfg: CLIPTextConfig) -> Spec:
    spec: Spec = []
    t = "text_model"
    spec.append((f"{t}.embeddings.token_embedding.weight", (cfg.vocab, cfg.width), "e"))
    spec.append((f"{t}.embeddings.position_embedding.weight", (cfg.max_len, cfg.width), "e"))
    for i in range(cfg.layers):
        p = f"{t}.encoder.layers.{i}"
        _norm(spec, p + ".layer_norm1", cfg.width)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _lin(spec, f"{p}.self_attn.{n}", cfg.width, cfg.width)
        _norm(spec, p + ".layer_norm2", cfg.width)
        _lin(spec, p + ".mlp.fc1", cfg.mlp, cfg.width)
        _lin(spec, p + ".mlp.fc2", cfg.width, cfg.mlp)
    _norm(spec, f"{t}.final_layer_norm", cfg.width)
    return spec


def count_params(spec: Spec) -> int:
    n = 0
    for _, shape, _ in spec:
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


_STRESS_RESIDUAL = (".conv2.weight", ".conv_shortcut.weight", ".to_out.0.weight", ".ff.net.2.weight", ".proj_out.weight", ".conv.weight",
                    "conv_in.weight", "conv_out.weight")


def _stress_class(name: str, level: int = 2):
    """(decades of per-channel scale, outlier factor) of a weight tensor in the range-stress set: see `synthesize`"""
    if name.endswith(".ff.net.0.proj.weight"):
        return (1.0, 1.0) if level >= 2 else (0.6, 1.0)   # GEGLU: the two halves multiply
    if (name.endswith(_STRESS_RESIDUAL) or name.startswith(("controlnet_", "time_embedding.", "add_embedding.")) or ".time_emb_proj." in name):
        return (1.0, 8.0) if level >= 2 else (0.6, 4.0)   # reads / writes the un-normalised residual stream (or the time path)
    return (2.0, 30.0) if level >= 2 else (1.0, 8.0)      # between two normalisations


def synthesize(spec: Spec, prefix: str = "", device="cpu", dtype=torch.float16, stress=False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: w ~ N(0, 1/fan_in), bias ~ 0.05 N, norm gamma ~ 1 + 0.1 N.

    Every tensor has its own generator seeded by crc32(prefix + name), so any subset can be
    re-created independently.  Values are produced in fp32 and rounded once to `dtype`; the oracle
    up-casts the same rounded values, so both sides compute with identical parameters.

    stress=True: the RANGE of a trained checkpoint instead of N(0, 1/fan_in) everywhere (no checkpoint exists
    offline).  Trained SD1.5 tensors differ from the plain synthetic ones in exactly the ways fp16 storage is sensitive to:
      * conv / linear weights get per-OUTPUT-channel scales spread log-uniformly (renormalised to unit mean square so that the
        layer's typical gain stays what the plain set has) and 0.5 % of the output channels (at least one) are outlier channels,
        larger still -- the massive-activation channels in front of the GroupNorms and on the residual stream.  How far depends on
        where the layer's output goes (`_stress_class`): layers between two normalisations (ResnetBlock conv1, q / k / v, proj_in)
        take two decades (0.1 ... 10) and 30 x outliers; layers that read or write the UN-normalised residual stream (conv2,
        out-projections, ff.net.2, proj_out, shortcuts, down / upsamplers, conv_in, the ControlNet's zero-convs and conditioning
        stack, the time path) one decade (0.3 ... 3) and 8 x outliers; the GEGLU projection (whose two halves MULTIPLY) one decade
        and no outliers.  Calibrated so that an fp16 run survives, as the reference's own fp16 pipeline must on a real checkpoint
        (the first calibration -- two decades and 30 x everywhere -- put 12 280 on the residual stream after one level and overflowed
        fp16 in the next shortcut conv, scripts/find_nonfinite.py: no fp16 reference would have survived it either);
      * norm weights: gamma log-uniform in 0.25 ... 4, beta ~ 0.3 N (away from (1, 0));
      * biases ~ 0.3 N;
      * attention logits pushed out: to_q / to_k 2.8 x larger each, so that the logit standard deviation is ~8 and a row of
        4 096 keys reaches +-30 (peaked softmax rows, as trained attention has).
    The extra draws come from a second generator per tensor (seed crc32(prefix + name + "#stress")), the plain values are the
    same as without the flag.
    stress=1: the milder set -- half the decades, 8 x / 4 x outliers, gamma in 0.5 ... 2, attention logits NOT pushed out.  fp16
    STORAGE alone (oracle.nets.EMULATE_FP16 against the fp32 oracle, full size) moves the denoised latents of the level-2 set by 27 %
    (peaked softmax rows turn rounding differences into different attention) and those of the level-1 set far less: the parity test
    holds the HIP path to that yardstick on both (tests/test_pipeline_gpu.py, the range-stress cases)."""
    out = {}
    dev = torch.device(device)
    for name, shape, kind in spec:
        g = torch.Generator(device=dev)
        g.manual_seed(zlib.crc32((prefix + name).encode()))
        x = torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        if kind == "w":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            x *= fan_in ** -0.5
        elif kind == "b":
            x *= 0.05
        elif kind == "g":
            x = 1.0 + 0.1 * x
        elif kind == "e":
            x *= 0.05
        level = 2 if stress is True else int(stress)
        if level:
            g2 = torch.Generator(device=dev)
            g2.manual_seed(zlib.crc32((prefix + name + "#stress").encode()))
            if kind == "w":
                n = shape[0]
                decades, outlier = _stress_class(name, level)
                sc = torch.pow(10.0, (torch.rand(n, generator=g2, device=dev) * 2.0 - 1.0) * decades / 2.0)  # log-uniform over `decades`
                sc = sc / sc.pow(2).mean().sqrt()
                n_out = max(1, int(round(0.005 * n))) if (n >= 64 and outlier > 1.0) else 0   # (not on 3- / 4-channel image / latent outputs)
                if n_out:
                    idx = torch.randperm(n, generator=g2, device=dev)[:n_out]
                    sc[idx] = sc[idx] * outlier
                x = x * sc.reshape((n,) + (1,) * (len(shape) - 1))
                if level >= 2 and (name.endswith(".to_q.weight") or name.endswith(".to_k.weight")):
                    x = x * 2.8
            elif kind == "b":
                x = x * 6.0                                                                  # 0.3 N
            elif kind == "g":
                x = torch.pow(4.0 if level >= 2 else 2.0, torch.rand(shape, generator=g2, device=dev) * 2.0 - 1.0)  # 0.25 ... 4 (0.5 ... 2)
                # (the matching beta is the "b" tensor that follows: 0.3 N)
        out[name] = x.to(dtype)
    return out


def load_safetensors(path: str, device="cpu", dtype=torch.float16) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file

    return {k: v.to(device=device, dtype=dtype) for k, v in load_file(path).items()}
