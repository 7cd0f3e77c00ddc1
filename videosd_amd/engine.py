"""Per-frame denoising engine: sequences the libvsd kernels for TAESD-encode -> N x (ControlNet, UNet,
LCM step) -> TAESD-decode, records the sequence once as a static program, and replays it as ONE hipGraph
per frame.

Replaces the body of LatentConsistencyModelPipeline_controlnet.__call__
(/root/reference/diffusert/lcm/lcm_controlnet.py:379-618) and the diffusers modules it drives.  All
compute goes through an `ops` object (videosd_amd.ops.HipOps -> libvsd.so); this module only decides
which buffers each kernel reads and writes.

Data layout in HBM
  * activations: fp16 [H*W][C] (channels-last), bump-allocated from one arena that is rewound at every
    denoising step so all steps reuse the same addresses;
  * 4-channel latent-like tensors: row stride 8, channels 4..7 zero;
  * weights: fp16 [N][Kp], K ordered (ky,kx,c) (videosd_amd/packing.py), resident for the process lifetime;
  * per-(strength,steps) constants: time-embedding projections of every ResnetBlock for every step, the
    noise draws, scheduler coefficients; per-prompt constants: cross-attention K and V^T of every layer.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from .config import ControlNetConfig, TAESDConfig, UNetConfig
from .lcm import LCMSchedule, timestep_sinusoid, w_embedding
from .ops import Geom
from .packing import PackedConv, add_frag, pack_conv, pack_geglu_ln, pack_linear, pack_linear_cat, pack_linear_ln
from .weights import skip_channels


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Arena:
    """Bump allocator over large device chunks; `mark`/`rewind` give every denoising step the same addresses."""

    def __init__(self, ops, chunk_bytes: int = 256 << 20):
        self.ops = ops
        self.chunk_bytes = chunk_bytes
        self.chunks: List[torch.Tensor] = []
        self.ci = 0
        self.off = 0
        self.peak = 0

    def alloc(self, rows: int, cols: int, dtype=torch.float16) -> torch.Tensor:
        esz = torch.empty(0, dtype=dtype).element_size()
        nbytes = _ru(rows * cols * esz, 256)
        if getattr(self.ops, "allocator", None) is not None:  # debugging hook (ops.HipOps.allocator): tensor by tensor
            self.peak += nbytes
            return self.ops.empty(rows, cols, dtype=dtype)
        while True:
            if self.ci >= len(self.chunks):  # (a tensor larger than the usual chunk gets a chunk of its own size)
                self.chunks.append(self.ops.empty(max(self.chunk_bytes, nbytes), dtype=torch.uint8))
                self.off = 0
            if self.off + nbytes <= self.chunks[self.ci].numel():
                break
            self.ci += 1
            self.off = 0
        t = self.chunks[self.ci][self.off:self.off + rows * cols * esz].view(dtype).view(rows, cols)
        self.off += nbytes
        self.peak = max(self.peak, sum(c.numel() for c in self.chunks[:self.ci]) + self.off)
        return t

    def mark(self) -> Tuple[int, int]:
        return (self.ci, self.off)

    def rewind(self, m: Tuple[int, int]):
        self.ci, self.off = m


class Recorder:
    """Records op calls as a static program; `run` replays them on the real ops object."""

    def __init__(self, ops):
        self.ops = ops
        self.calls = []

    def __getattr__(self, name):
        fn = getattr(self.ops, name)

        def rec(*a, **k):
            self.calls.append((fn, a, k))

        return rec

    def run(self):
        for fn, a, k in self.calls:
            fn(*a, **k)

    # a stretch of the program recorded in two FORMS (same buffers, same results): `flavor(i)` is the program with form i
    def variants(self, form0, form1):
        self.calls.append((_variants, (form0, form1), {}))

    def flavor(self, i: int) -> "Recorder":
        out = Recorder(self.ops)
        for c in self.calls:
            if c[0] is _variants:
                out.calls += c[1][i]
            else:
                out.calls.append(c)
        return out


def _variants(*a, **k):
    raise RuntimeError("a program with variants is run through Recorder.flavor")


# ------------------------------------------------------------------------------------------ packed weights
@dataclass
class ResnetW:
    cin: int
    cout: int
    n1: Tuple[torch.Tensor, torch.Tensor]
    conv1: PackedConv          # bias folded into the time projection
    n2: Tuple[torch.Tensor, torch.Tensor]
    conv2: PackedConv
    shortcut: Optional[PackedConv]
    temb_off: int              # column offset into the concatenated time-projection output


@dataclass
class BlockW:
    """One BasicTransformerBlock."""
    qkv: PackedConv            # LayerNorm norm1 folded in (pack_linear_ln)
    out1: PackedConv
    q2: PackedConv             # norm2 folded in
    kv2: PackedConv
    out2: PackedConv
    ff1: PackedConv            # norm3 folded in, GEGLU tile-packed
    ff2: PackedConv
    kv_index: int              # index of this block's entries in a prompt's constants (PromptLayout)
    # "absorbed" cross-attention (C >= XATTN_ABSORB_MIN_C): the text's key / value projections folded into the query and
    # output weights, rebuilt per prompt on the GPU (vsd_xattn_fold); what that needs besides out2: device copies of
    xa_raw: Optional[tuple] = None     # (raw to_q weight fp16 [C][C], norm2 gamma, norm2 beta)


@dataclass
class TransformerW:
    """Transformer2DModel: GroupNorm, proj_in, `depth` blocks (1 for SD1.5; 2 / 10 for SDXL), proj_out."""
    c: int
    norm: Tuple[torch.Tensor, torch.Tensor]
    proj_in: PackedConv
    blocks: List[BlockW]
    proj_out: PackedConv


# Cross-attention as two GEMMs pays when one 128-column tile per head is not wider than the query projection it
# replaces: 8 heads * 128 = 1024 columns against C (SD1.5: the 640- and 1280-wide levels; see pack_cross_attention)
XATTN_ABSORB_MIN_C = 640


class NetWeights:
    """Device-resident, kernel-layout weights of one UNet-shaped network (UNet or ControlNet encoder)."""

    def __init__(self, ops, cfg: UNetConfig, w: Dict[str, torch.Tensor], is_controlnet=False,
                 cn_cfg: Optional[ControlNetConfig] = None):
        self.ops, self.cfg, self.is_cn = ops, cfg, is_controlnet
        self._w = w
        self._temb_w, self._temb_b = [], []
        self._temb_cols = 0
        self.transformers: List[BlockW] = []  # every BasicTransformerBlock, in kv_cache order
        dev = ops.to_device
        ch = cfg.block_out_channels
        self.conv_in = self._conv("conv_in", cin_pad=8)
        self.time_l1 = self._lin("time_embedding.linear_1")
        self.time_l2 = self._lin("time_embedding.linear_2")
        self.cond_proj = self._lin("time_embedding.cond_proj") if cfg.cond_proj_dim else None
        self.add_l1 = self._lin("add_embedding.linear_1") if cfg.add_time_dim else None
        self.add_l2 = self._lin("add_embedding.linear_2") if cfg.add_time_dim else None
        self.down: List[List[Tuple[ResnetW, Optional[TransformerW]]]] = []
        self.downsamplers: List[Optional[PackedConv]] = []
        cin = ch[0]
        for i, cout in enumerate(ch):
            blk = []
            for j in range(cfg.layers_per_block):
                r = self._resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
                t = (self._transformer(f"down_blocks.{i}.attentions.{j}", cout, cfg.transformer_depth[i])
                     if cfg.down_attn[i] else None)
                blk.append((r, t))
            self.down.append(blk)
            self.downsamplers.append(self._conv(f"down_blocks.{i}.downsamplers.0.conv") if i < len(ch) - 1 else None)
            cin = cout
        self.mid = (self._resnet("mid_block.resnets.0", ch[-1], ch[-1]),
                    self._transformer("mid_block.attentions.0", ch[-1], cfg.mid_depth),
                    self._resnet("mid_block.resnets.1", ch[-1], ch[-1]))
        if not is_controlnet:
            skips = skip_channels(cfg)
            rev = list(reversed(ch))
            self.up: List[List[Tuple[ResnetW, Optional[TransformerW]]]] = []
            self.upsamplers: List[Optional[PackedConv]] = []
            prev = ch[-1]
            for i, cout in enumerate(rev):
                blk = []
                for j in range(cfg.layers_per_block + 1):
                    sc = skips.pop()
                    r = self._resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else cout) + sc, cout)
                    t = (self._transformer(f"up_blocks.{i}.attentions.{j}", cout, cfg.up_depth[i])
                         if cfg.up_attn[i] else None)
                    blk.append((r, t))
                self.up.append(blk)
                self.upsamplers.append(self._conv(f"up_blocks.{i}.upsamplers.0.conv") if i < len(rev) - 1 else None)
                prev = cout
            self.norm_out = self._norm("conv_norm_out")
            self.conv_out = self._conv("conv_out")
        else:
            cc = cn_cfg.cond_channels
            p = "controlnet_cond_embedding"
            self.cond_convs = [(self._conv(f"{p}.conv_in", cin_pad=8), 1)]
            k = 0
            for i in range(len(cc) - 1):
                self.cond_convs.append((self._conv(f"{p}.blocks.{k}"), 1))
                self.cond_convs.append((self._conv(f"{p}.blocks.{k + 1}"), 2))
                k += 2
            self.cond_out = self._conv(f"{p}.conv_out")
            self.zero_convs = [self._conv(f"controlnet_down_blocks.{i}") for i in range(len(skip_channels(cfg)))]
            self.zero_mid = self._conv("controlnet_mid_block")
        # all per-ResnetBlock time projections as ONE linear layer: [sum Cout][temb_dim]
        tw = pack_linear_cat(self._temb_w, self._temb_b)
        self.temb_proj = self._to_dev(tw)
        self._w = None
        self._temb_w = self._temb_b = None

    # --- helpers
    def _to_dev(self, p: PackedConv) -> PackedConv:
        for f in ("weight", "bias", "ln_s", "ln_t", "weight_frag"):
            v = getattr(p, f)
            if v is not None:
                setattr(p, f, self.ops.to_device(v.contiguous()))
        return p

    def _conv(self, name, cin_pad=None, with_bias=True) -> PackedConv:
        b = self._w.get(name + ".bias") if with_bias else None
        return self._to_dev(pack_conv(self._w[name + ".weight"], b, cin_pad=cin_pad))

    def _lin(self, name) -> PackedConv:
        return self._to_dev(pack_linear(self._w[name + ".weight"], self._w.get(name + ".bias")))

    def _norm(self, name):
        return (self.ops.to_device(self._w[name + ".weight"].half().contiguous()),
                self.ops.to_device(self._w[name + ".bias"].half().contiguous()))

    def _resnet(self, p, cin, cout) -> ResnetW:
        off = self._temb_cols
        self._temb_w.append(self._w[p + ".time_emb_proj.weight"])
        # conv1's bias is folded into the time projection's bias: both are per-channel constants added
        # to conv1's output before norm2 (ResnetBlock2D.forward).
        self._temb_b.append((self._w[p + ".time_emb_proj.bias"].float() + self._w[p + ".conv1.bias"].float()))
        self._temb_cols += cout
        sc = self._conv(p + ".conv_shortcut") if (p + ".conv_shortcut.weight") in self._w else None
        return ResnetW(cin, cout, self._norm(p + ".norm1"), self._conv(p + ".conv1", with_bias=False),
                       self._norm(p + ".norm2"), self._conv(p + ".conv2"), sc, off)

    def _transformer(self, p, c, depth=1) -> TransformerW:
        w = self._w
        blocks = []
        for kb in range(depth):
            b = f"{p}.transformer_blocks.{kb}"
            ln = lambda n: (w[f"{b}.{n}.weight"], w[f"{b}.{n}.bias"])  # noqa: E731
            qkv = self._to_dev(pack_linear_ln([w[f"{b}.attn1.to_q.weight"], w[f"{b}.attn1.to_k.weight"],
                                               w[f"{b}.attn1.to_v.weight"]], None, *ln("norm1")))
            q2 = self._to_dev(pack_linear_ln([w[f"{b}.attn2.to_q.weight"]], None, *ln("norm2")))
            kv2 = self._to_dev(pack_linear_cat([w[f"{b}.attn2.to_k.weight"], w[f"{b}.attn2.to_v.weight"]]))
            ff1 = self._to_dev(pack_geglu_ln(w[f"{b}.ff.net.0.proj.weight"], w[f"{b}.ff.net.0.proj.bias"], *ln("norm3")))
            blk = BlockW(qkv, self._lin(b + ".attn1.to_out.0"), q2, kv2, self._lin(b + ".attn2.to_out.0"), ff1,
                         self._lin(b + ".ff.net.2"), len(self.transformers))
            heads = self.cfg.heads_for(c)
            if c >= XATTN_ABSORB_MIN_C and c % heads == 0 and c % 64 == 0 and (c // heads) % 8 == 0:
                dv = lambda t: self.ops.to_device(t.detach().to(torch.float16).contiguous())  # noqa: E731
                blk.xa_raw = (dv(w[f"{b}.attn2.to_q.weight"]), dv(ln("norm2")[0]), dv(ln("norm2")[1]))
            self.transformers.append(blk)
            blocks.append(blk)
        # use_linear_projection (SDXL): Linear on the token matrix == the 1x1 conv of SD1.5 in this layout
        proj = self._lin if self.cfg.linear_proj else self._conv
        tw = TransformerW(c, self._norm(p + ".norm"), proj(p + ".proj_in"), blocks, proj(p + ".proj_out"))
        if c == getattr(self.ops, "TAIL_C", 0) and depth == 1:
            # the fused per-token chains (csrc/fused_tail.hip) read these six matrices fragment-major
            for pc in (blocks[0].out1, blocks[0].q2, blocks[0].out2, blocks[0].ff1, blocks[0].ff2, tw.proj_out):
                pc.weight_frag = self.ops.to_device(add_frag(PackedConv(pc.weight.cpu(), None, pc.n, pc.k, pc.kp, pc.cin, pc.ksize)).weight_frag)
        return tw


class TAESDWeights:
    def __init__(self, ops, w: Dict[str, torch.Tensor]):
        self.ops = ops

        def cv(name, cin_pad=None):
            p = pack_conv(w[name + ".weight"], w.get(name + ".bias"), cin_pad=cin_pad)
            p.weight = ops.to_device(p.weight)
            if p.bias is not None:
                p.bias = ops.to_device(p.bias)
            return p

        def blk(p):
            return [cv(f"{p}.conv.{k}") for k in (0, 2, 4)]

        e = "encoder.layers"
        self.enc_in = cv(f"{e}.0", cin_pad=8)
        self.enc_blocks0 = [blk(f"{e}.1")]
        self.enc_stages = []
        n = 2
        for _ in range(3):
            down = cv(f"{e}.{n}")
            n += 1
            bs = []
            for _ in range(3):
                bs.append(blk(f"{e}.{n}"))
                n += 1
            self.enc_stages.append((down, bs))
        self.enc_out = cv(f"{e}.{n}")
        d = "decoder.layers"
        self.dec_in = cv(f"{d}.0", cin_pad=8)
        self.dec_stages = []
        n = 2
        for nb in (3, 3, 3):
            bs = []
            for _ in range(nb):
                bs.append(blk(f"{d}.{n}"))
                n += 1
            n += 1
            up = cv(f"{d}.{n}")
            n += 1
            self.dec_stages.append((bs, up))
        self.dec_last_block = blk(f"{d}.{n}")
        n += 1
        self.dec_out = cv(f"{d}.{n}")


class PromptLayout:
    """Byte layout of ONE prompt's device constants for a family of engines (same weights): per BasicTransformerBlock of the
    UNet / ControlNet its cross-attention K [tl][C] and V^T [C][ldt], and for the wide blocks the "absorbed" query / output
    weights (xa1: weights [heads*128][C] + ln_s / ln_t, xa2: weights [C][heads*128] + bias).  Everything lives in ONE
    contiguous buffer (~40 MB for SD1.5 + ControlNet), so that switching an engine to another cached prompt is a single
    device-to-device copy and a prompt cache entry is a single allocation."""

    def __init__(self, nets, tl: int):
        self.tl, self.ldt = tl, _ru(tl, 64)
        self.items = {}   # (net index, block index, name) -> (offset, shape, dtype)
        self.absorbed = set()
        off = 0

        def add(key, shape, dtype):
            nonlocal off
            n = 1
            for d in shape:
                n *= d
            self.items[key] = (off, tuple(shape), dtype)
            off = _ru(off + n * torch.empty(0, dtype=dtype).element_size(), 256)

        for ni, net in enumerate(nets):
            for bi, t in enumerate(net.transformers):
                c = t.kv2.n // 2
                add((ni, bi, "k"), (tl, c), torch.float16)
                add((ni, bi, "vt"), (c, self.ldt), torch.float16)
                if t.xa_raw is not None and tl <= 128:
                    hg = net.cfg.heads_for(c) * 128
                    self.absorbed.add((ni, bi))
                    add((ni, bi, "xa1_w"), (hg, c), torch.float16)
                    add((ni, bi, "xa1_s"), (hg,), torch.float32)
                    add((ni, bi, "xa1_t"), (hg,), torch.float32)
                    add((ni, bi, "xa2_w"), (c, hg), torch.float16)
                    add((ni, bi, "xa2_b"), (c,), torch.float16)
        self.nbytes = max(off, 256)


class PromptBlock:
    """One prompt's constants in device memory (layout: PromptLayout).  Engines read THEIR OWN block (its addresses are in
    their captured graphs); a cached prompt is installed by copying its block over the engine's."""

    def __init__(self, ops, layout: PromptLayout):
        self.layout = layout
        self.buf = ops.zeros(layout.nbytes, dtype=torch.uint8)  # zero: V^T key padding and the unused rows of the xa weights
        self._xa = {}
        self.text = None

    def view(self, ni, bi, name):
        off, shape, dtype = self.layout.items[(ni, bi, name)]
        n = 1
        for d in shape:
            n *= d
        return self.buf[off:off + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(*shape)

    def kv(self, ni, bi):
        return self.view(ni, bi, "k"), self.view(ni, bi, "vt")

    def xa(self, ni, bi):
        """(xa1, xa2) PackedConv pair of an absorbed block (views into this block), or None"""
        if (ni, bi) not in self.layout.absorbed:
            return None
        got = self._xa.get((ni, bi))
        if got is None:
            w1, w2 = self.view(ni, bi, "xa1_w"), self.view(ni, bi, "xa2_w")
            hg, c = w1.shape
            x1 = PackedConv(w1, None, hg, c, c, c, 1, ln_s=self.view(ni, bi, "xa1_s"), ln_t=self.view(ni, bi, "xa1_t"), tile128=True)
            x2 = PackedConv(w2, self.view(ni, bi, "xa2_b"), c, hg, hg, hg, 1)
            got = self._xa[(ni, bi)] = (x1, x2)
        return got


class RefCtx:
    """Bookkeeping of the reference-only mode while the program is recorded (SURVEY.md 8f-4;
    /root/reference/diffusert/lcm/lcm_reference_pipeline.py:498-794).  mode "write": the UNet pass over the noised reference
    latents leaves, per BasicTransformerBlock, its self-attention K rows and V^T columns in the SECOND half of buffers the
    read pass fills the first half of (the bank of norm1 outputs, :527, enters attention only through to_k / to_v, and
    those projections are exactly the write pass's own K / V), and per gated block output its per-channel (sum, sumsq)
    (:587-598).  mode "read": self-attention runs over [x ; bank] keys (:535-539) and the gated outputs are AdaIN-ed to the
    banked statistics (:593-603)."""

    def __init__(self, cfg):
        self.mode = "write"
        self.kv = {}      # kv_index -> (qk_full [2*hw][2c], vt_full [c][ldvt])
        self.stats = {}   # (kind, i, j) -> fp32 [c][2] of the write pass
        nd = nu = len(cfg.block_out_channels)
        self.nd, self.nu = nd, nu

    def gate(self, kind, i, weight=1.0):
        # module.gn_weight, lcm_reference_pipeline.py:816-846 (gn_auto_machine_weight = 1.0)
        gw = 0.0 if kind == "mid" else (2.0 * (1.0 - i / self.nd) if kind == "down" else 2.0 * i / self.nu)
        return weight >= gw


# ------------------------------------------------------------------------------------------ the engine
class Engine:
    """One engine = one GPU, one weight replica (reference: one Ray actor, videopipeline.py:11-32)."""

    def __init__(self, ops, unet_cfg: UNetConfig, cn_cfg: ControlNetConfig, vae_cfg: TAESDConfig,
                 w_unet: Dict[str, torch.Tensor], w_cn: Optional[Dict[str, torch.Tensor]], w_vae: Dict[str, torch.Tensor],
                 guidance_scale: float = 7.5):
        self.ops = ops
        self.ucfg, self.ccfg, self.vcfg = unet_cfg, cn_cfg, vae_cfg
        self.unet = NetWeights(ops, unet_cfg, w_unet)
        self.cn = NetWeights(ops, cn_cfg.unet, w_cn, True, cn_cfg) if w_cn is not None else None
        self.vae = TAESDWeights(ops, w_vae)
        self.guidance_scale = guidance_scale  # never forwarded by the reference (videopipeline.py:114-124): 7.5
        self.text = None
        self.added = None  # SDXL: (pooled text embeds, 6 time ids)
        self.plan = None
        self.graph = None
        self.program = self.program_serial = None
        self.use_graph = True
        self.is_slot = False
        self.batch = 1
        self.overlap_controlnet = True  # record the ControlNet encoder for a second stream, parallel to the UNet encoder
        self.overlap_launch = True      # ... and launch it that way by default (`launch(overlap=...)` decides per launch)
        # kernel choices of the plan: False = each layer's fastest form ALONE on an idle GPU (a lone launch: latency), True = the
        # form that costs least with four launch lanes busy (throughput; ops.HipOps.tune_conv).  Fixed per plan at `prepare`.
        self.tune_for_lanes = False
        self.graph_serial = None
        # ... and, optionally, while that stream is otherwise idle: the ControlNet skip merges beside the mid block / decoder
        # start and Sobel + conditioning embedding beside the TAESD encoder (`use_side_stream`).  Measured on MI355X (512x512,
        # 4 steps): one frame alone 23.8 -> 23.2 ms, but with two launches in flight 97 -> 81 frames/s -- four busy hardware
        # queues instead of two cost more than the filled gaps give.  Round 4, launch streams on their own pipes: level either
        # way (one frame 47.8 vs 47.6 launches/s, 5 x 2 128.0 vs 128.4).  Off; `VSD_SIDE=1` switches it on.  (Round 2 also
        # moved the decoder's 1x1 shortcut convs there: 23.6 ms, removed.)
        self.use_side_stream = False
        import os as _os
        if _os.environ.get("VSD_SIDE") is not None:  # A/B switch
            self.use_side_stream = _os.environ.get("VSD_SIDE") == "1"
        if _os.environ.get("VSD_OVERLAP_CN") is not None:  # A/B switch (callers that set the attribute afterwards still win)
            self.overlap_controlnet = _os.environ.get("VSD_OVERLAP_CN") == "1"
        self._ev_count = 0
        # what a slot shares with its parent besides the weights: the per-plan constant block (schedule coefficients,
        # ControlNet scales) that the captured graphs READ, so that `update_options` reaches every graph at once
        self.shared = {}
        # what EVERY engine made from this one shares (weights aside): the prompt-constant layout and the default prompt
        # (`set_text_embeds`); an engine follows the default unless `use_prompt` gave it one of its own
        self.family = {}
        self.pblock = None      # this engine's own prompt constants (the addresses its program / graph reads)
        self._installed = None  # the PromptBlock whose bytes `pblock` currently holds
        self._want = None       # `use_prompt`: this engine's prompt (None: the family default)
        self.absorb_cross_attention = True  # cross-attention of the wide blocks as two GEMMs (vsd_xattn_fold)
        self.use_fused_tail = True          # 320-wide blocks: per-token chains as fused launches (csrc/fused_tail.hip)
        self.group_merges = not __import__("os").environ.get("VSD_NO_GROUP")  # the ControlNet merges of a step as two grouped launches
        # the one-stream form of the program runs the UNet and the ControlNet encoder of a step in lock step, twin layers as one
        # grid (`prepare`: both forms are recorded, `launch` picks)
        self.twin_encoders = not __import__("os").environ.get("VSD_NO_TWIN")
        self.group_shortcuts = not __import__("os").environ.get("VSD_NO_GROUP_SHORTCUT")  # (a ResnetBlock's shortcut conv in conv1's grid)
        self.tail_b_min_rows = 1024         # ... the feed-forward chain from this many tokens per launch on (see _transformer)

    def make_slot(self, share_plan: bool = True, lane: Optional[int] = None) -> "Engine":
        """A further engine on the same GPU and the same weights, with its own streams, arena, I/O buffers, prompt constants
        and graph.
        share_plan=True: a further frame in flight of THIS engine's plan -- shares its schedule constants (`update_options`
        on the parent reaches it).  Call `prepare` on the parent first, then on the slot with the same arguments.  (The
        reference keeps one frame in flight per Ray actor; frames are independent, so a second one fills the gaps the
        first leaves between its small dependent kernels.)
        share_plan=False: an independent plan (another frame size / step count / session) with constants of its own: it is
        prepared and updated like a parent, while the other plans keep running (server.py:90-93: options are per session,
        every session's frames go through the same actors).
        lane: the launch lane whose streams the engine uses (HipOps.clone; default: the next unused lane).  Engines of one lane
        take turns on its streams, engines of different lanes run side by side."""
        e = Engine.__new__(Engine)
        e.__dict__.update(self.__dict__)
        e.ops = self.ops.clone(lane)
        e._vt_pool = {}
        e.graph = None
        e.graph_serial = None
        e.plan = None
        e._stage = None  # own pinned staging buffers and events (a copy of the parent's would be SHARED with it)
        e.pblock = None
        e._installed = None
        e._want = None
        e.is_slot = bool(share_plan)
        if not share_plan:
            e.shared = {}
        return e

    # ---------------------------------------------------------------- prompt-dependent constants
    def _nets(self):
        return [self.unet] + ([self.cn] if self.cn else [])

    def build_prompt(self, embeds: torch.Tensor) -> PromptBlock:
        """embeds: [77, cross_dim] (or [1,77,cross_dim]) -> a fresh PromptBlock: cross-attention K and V^T of every layer (the
        to_k / to_v projections diffusers recomputes every step, attn2 of each block) and, for the wide blocks, the query /
        output weights with K / V folded in (vsd_xattn_fold) -- all on the GPU (~1 ms), touching nothing a running launch
        reads: a prompt cache is filled while frames of other prompts are in flight."""
        ops = self.ops
        e = embeds.reshape(-1, embeds.shape[-1]).to(torch.float16).contiguous()
        text = ops.to_device(e)
        tl = text.shape[0]
        # one layout OBJECT per text length for the whole family: prepared plans and cached prompt blocks compare layouts by
        # identity, so a 77 -> other -> 77 sequence of prompt lengths must come back to the first object
        lays = self.family.setdefault("layouts", {})
        lay = lays.get(tl)
        if lay is None:
            lay = lays[tl] = PromptLayout(self._nets(), tl)
        self.family["layout"] = lay
        blk = PromptBlock(ops, lay)
        blk.text = text
        for ni, net in enumerate(self._nets()):
            for bi, t in enumerate(net.transformers):
                c = t.kv2.n // 2
                k, vt = blk.kv(ni, bi)
                ops.conv(text, None, Geom.linear(tl), t.kv2, k, ldo=c, out_t=vt, ldt=lay.ldt, t_col0=c)
                if (ni, bi) in lay.absorbed:
                    wq, ga, be = t.xa_raw
                    heads = net.cfg.heads_for(c)
                    ops.xattn_fold(k, vt, tl, wq, t.out2.weight, ga, be, c, heads, (c // heads) ** -0.5, blk.view(ni, bi, "xa1_w"),
                                   blk.view(ni, bi, "xa1_s"), blk.view(ni, bi, "xa1_t"), blk.view(ni, bi, "xa2_w"))
                    ops.copy_(blk.view(ni, bi, "xa2_b"), t.out2.bias)
        ops.synchronize()
        return blk

    def set_text_embeds(self, embeds: torch.Tensor):
        """The prompt of this engine AND of every engine made from it that was not given one of its own (`use_prompt`).
        Prepared plans stay valid: an engine copies the new constants over its own block before its next launch."""
        blk = self.build_prompt(embeds)
        self.family["prompt"] = blk
        self._want = None
        self.text = blk.text

    def use_prompt(self, blk: Optional[PromptBlock]):
        """This engine's launches use `blk` (from `build_prompt`, e.g. a prompt cache entry) from the next launch on, whatever
        the other engines of the family run; None: follow the family default again.  Call it with no launch of THIS engine
        in flight (the other lanes are not affected: every engine reads its own copy)."""
        self._want = blk

    def _sync_prompt(self):
        src = self._want if self._want is not None else self.family.get("prompt")
        if src is None:
            raise RuntimeError("set_text_embeds (or use_prompt) must be called before a launch")
        if src is not self._installed:
            if self.pblock is None or src.layout is not self.pblock.layout:
                raise RuntimeError("the prompt's text length differs from the one this plan was prepared for: prepare again")
            self.ops.copy_(self.pblock.buf, src.buf)  # one device-to-device copy on this engine's own stream
            self._installed = src

    def set_added_cond(self, pooled: torch.Tensor, time_ids):
        """SDXL micro-conditioning: pooled text embedding [add_pooled_dim] and the 6 time ids
        (orig_h, orig_w, crop_top, crop_left, target_h, target_w).  Takes effect at the next `prepare`."""
        self.added = (pooled.detach().float().cpu().reshape(-1), [float(v) for v in time_ids])

    # ---------------------------------------------------------------- schedule-dependent constants
    def _time_embeddings(self, net: NetWeights, sched: LCMSchedule, out: torch.Tensor, use_cond: bool = True):
        """out[i] = concat over ResnetBlocks of time_emb_proj(SiLU(time_embedding(t_i))) + conv1.bias.
        Replaces Timesteps/TimestepEmbedding + every ResnetBlock2D.time_emb_proj (K9 in SURVEY.md)."""
        ops, cfg = self.ops, net.cfg
        n = len(sched)
        c0 = cfg.block_out_channels[0]
        t_emb = ops.to_device(timestep_sinusoid(sched.timesteps, c0).half())
        x = t_emb
        if net.cond_proj is not None and use_cond:  # (the reference-only WRITE pass passes no timestep_cond)
            wemb = ops.to_device(w_embedding(self.guidance_scale, cfg.cond_proj_dim).half().expand(n, -1).contiguous())
            x = ops.empty(n, c0)
            ops.conv(wemb, None, Geom.linear(n), net.cond_proj, x, residual=t_emb)
        h1 = ops.empty(n, cfg.temb_dim)
        ops.conv(x, None, Geom.linear(n), net.time_l1, h1, act=L.ACT_SILU)
        h2 = ops.empty(n, cfg.temb_dim)
        if net.add_l1 is not None:
            # SDXL text_time conditioning: temb += add_embedding(cat[pooled text embeds, sinusoids of the 6 time ids])
            # (UNet2DConditionModel.get_aug_embed); the same vector for every step
            if self.added is None:
                raise RuntimeError("this UNet needs set_added_cond(pooled, time_ids) before prepare")
            pooled, time_ids = self.added
            tid = timestep_sinusoid([float(v) for v in time_ids], cfg.add_time_dim).reshape(1, -1)
            a0 = ops.to_device(torch.cat([pooled.float().reshape(1, -1), tid], dim=-1).half().expand(n, -1).contiguous())
            a1 = ops.empty(n, cfg.temb_dim)
            ops.conv(a0, None, Geom.linear(n), net.add_l1, a1, act=L.ACT_SILU)
            aug = ops.empty(n, cfg.temb_dim)
            ops.conv(a1, None, Geom.linear(n), net.add_l2, aug)
            ops.conv(h1, None, Geom.linear(n), net.time_l2, h2, residual=aug, act=L.ACT_SILU | L.ACT_POST)
        else:
            ops.conv(h1, None, Geom.linear(n), net.time_l2, h2, act=L.ACT_SILU)  # SiLU(temb): all consumers apply it
        ops.conv(h2, None, Geom.linear(n), net.temb_proj, out[:n])
        ops.synchronize()

    def _temb(self, net):
        """this plan's per-step time-embedding projections of `net` (device constants shared by the plan's engines)"""
        return self.shared["temb"]["cn" if net is self.cn else "unet"]

    # ---------------------------------------------------------------- network builders (record ops)
    def _resnet(self, r, rw: ResnetW, net, step, x, x2, c0, c1, hw, geom, out=None, out2=None, add2=None, residual2=None,
                temb=None, stat_out=None):
        """x (and optional concat partner x2) -> ResnetBlock2D output [batch*hw][cout]  (hw = pixels per image).
        temb: another time-projection table than the plan's; stat_out: fp32 [cout][2] to receive the output's per-channel
        (sum, sumsq) (reference-only AdaIN)."""
        a, cfg = self.arena, net.cfg
        cin = c0 + c1
        rows = self.batch * hw
        t1 = a.alloc(rows, cin)
        self._gn(r, x, x2, c0, c1, hw, cfg.groups, 1e-5, rw.n1[0], rw.n1[1], True, t1)
        h = a.alloc(rows, rw.cout)
        tv = (self._temb(net) if temb is None else temb)[step, rw.temb_off:rw.temb_off + rw.cout]
        sc = x
        # The shortcut conv (ResnetBlock2D.conv_shortcut: 16 per denoising step with the ControlNet) depends on the block's INPUT only:
        # it shares conv1's grid (vsd_conv_gemm_group, every member at the split it has as a launch of its own: same bits) instead of
        # being one more dependent launch between norm2 and conv2 -- 64 launches of a 4-step frame less.
        grouped = (rw.shortcut is not None and self.group_shortcuts and hasattr(self.ops, "conv_group") and
                   cin % 64 == 0 and c0 % 64 == 0 and c1 % 64 == 0 and rw.cout % 64 == 0 and (geom.hi, geom.wi) == (geom.hs, geom.ws))
        if grouped:
            sc = a.alloc(rows, rw.cout)
            r.conv_group([((t1, None, geom, rw.conv1, h), dict(rowvec=tv)),
                          ((x, x2, Geom.linear(rows), rw.shortcut, sc), dict(c0=c0, c1=c1))], split="own")
        else:
            r.conv(t1, None, geom, rw.conv1, h, rowvec=tv)
        t2 = a.alloc(rows, rw.cout)
        self._gn(r, h, None, rw.cout, 0, hw, cfg.groups, 1e-5, rw.n2[0], rw.n2[1], True, t2)
        if rw.shortcut is not None and not grouped:
            sc = a.alloc(rows, rw.cout)
            r.conv(x, x2, Geom.linear(rows), rw.shortcut, sc, c0=c0, c1=c1)
        out = out if out is not None else a.alloc(rows, rw.cout)
        r.conv(t2, None, geom, rw.conv2, out, residual=sc, residual2=residual2, out2=out2, add2=add2,
               chanstat_out=stat_out)
        return out

    def _transformer(self, r, tw: TransformerW, net, x, hw, out2=None, add2=None, ref: Optional[RefCtx] = None, stat_out=None):
        a, cfg = self.arena, net.cfg
        c = tw.c
        heads = cfg.heads_for(c)
        d = c // heads
        B = self.batch
        rows = B * hw                      # the token matrix of all images in flight: [B*hw][C]
        lin = Geom.linear(rows)
        t = a.alloc(rows, c)
        self._gn(r, x, None, c, 0, hw, cfg.groups, 1e-6, tw.norm[0], tw.norm[1], False, t)
        # The three LayerNorms are never materialised: every producer of the token stream leaves per-row
        # (sum, sumsq) partials (rowstat_out) and the consuming GEMM applies the norm in its epilogue (ln_part).
        ng = c // 64
        stat = lambda: a.alloc(rows, ng * 2, dtype=torch.float32).view(rows, ng, 2)  # noqa: E731
        h = a.alloc(rows, c)
        rs = stat()
        r.conv(t, None, lin, tw.proj_in, h, rowstat_out=rs)
        t_img = _ru(hw, 64)                # V^T columns per image (key padding stays zero)
        ldvt = B * t_img
        for bi, bw in enumerate(tw.blocks):
            # self-attention (per image: keys never cross an image boundary)
            att = a.alloc(rows, c)
            if ref is None:
                qk = a.alloc(rows, 2 * c)
                vt = self._vt_buffer(c, ldvt)
                r.conv(h, None, Geom.linear(hw, batch=B), bw.qkv, qk, ldo=2 * c, out_t=vt, ldt=ldvt, t_col0=2 * c, ln_part=rs,
                       t_img=t_img)
                r.attention(qk, 2 * c, qk[:, c:], 2 * c, vt, ldvt, att, c, hw, hw, heads, d, d ** -0.5, batch=B, k_brows=hw,
                            vt_bcols=t_img)
            else:
                # reference-only: K rows / V^T columns [0, hw) belong to the frame (read pass), [hw, 2 hw) to the reference
                # (write pass); the read pass attends over all 2 hw keys (lcm_reference_pipeline.py:535-539)
                if ref.mode == "write":
                    # the write pass reads whole 64-key tiles from column hw on: the row must hold hw + ru(hw, 64) columns
                    # so that its masked tail keys are this buffer's zero padding, not the next row's read-pass columns
                    # (masked probabilities are 0, but 0 x a stale Inf/NaN is NaN in the PV product)
                    ld2 = _ru(hw + _ru(hw, 64), 64)
                    ref.kv[bw.kv_index] = (a.alloc(2 * hw, 2 * c), self._vt_buffer(c, ld2))
                qk_full, vt_full = ref.kv[bw.kv_index]
                ld2 = vt_full.shape[1]
                off = hw if ref.mode == "write" else 0
                qk = qk_full[off:off + hw]
                r.conv(h, None, Geom.linear(hw), bw.qkv, qk, ldo=2 * c, out_t=vt_full[:, off:], ldt=ld2, t_col0=2 * c, ln_part=rs)
                if ref.mode == "write":
                    r.attention(qk, 2 * c, qk[:, c:], 2 * c, vt_full[:, off:], ld2, att, c, hw, hw, heads, d, d ** -0.5)
                else:
                    r.attention(qk, 2 * c, qk_full[:, c:], 2 * c, vt_full, ld2, att, c, hw, 2 * hw, heads, d, d ** -0.5)
            ni = 1 if net is self.cn else 0
            kt, vtt = self.pblock.kv(ni, bw.kv_index)
            xa = self.pblock.xa(ni, bw.kv_index) if self.absorb_cross_attention else None
            fused = (self.use_fused_tail and c == getattr(self.ops, "TAIL_C", 0) and len(tw.blocks) == 1 and ref is None and
                     stat_out is None and out2 is None and hasattr(self.ops, "tail_a"))
            if fused:
                # the block's per-token chains as fused launches around the cross-attention (csrc/fused_tail.hip): the token
                # tile's owner streams only the weights; h1 / q (and in tail_b h2, the GEGLU hidden state, h3) stay on chip.
                # Measured on MI355X (us, fused vs the launches it replaces; the library picks the tile height per token count):
                # tail_a 10 vs 19 at 4096 tokens, 15 vs 31 at 12288, 21 vs 53 at 20480; tail_b 44 vs 59 at 4096 (every
                # workgroup streams all 2.9 MB of feed-forward weights whatever the token count), 61 vs 111 at 12288,
                # 91 vs 214 at 20480.  Below ~2048 tokens (small frames) the separate launches win.
                h1 = a.alloc(rows, c)
                q = a.alloc(rows, c)
                r.tail_a(att, h, rows, bw.out1, bw.q2, h1, q)
                r.attention(q, c, kt, c, vtt, vtt.shape[1], att, c, rows, kt.shape[0], heads, d, d ** -0.5)
                if rows >= self.tail_b_min_rows:
                    out = a.alloc(rows, c)
                    r.tail_b(att, h1, x, rows, bw.out2, bw.ff1, bw.ff2, tw.proj_out, out)
                    return out
                h2 = a.alloc(rows, c)
                rs2 = stat()
                r.conv(att, None, lin, bw.out2, h2, residual=h1, rowstat_out=rs2)
                f = a.alloc(rows, 4 * c)
                r.conv(h2, None, lin, bw.ff1, f, ln_part=rs2)
                h3 = a.alloc(rows, c)
                r.conv(f, None, lin, bw.ff2, h3, residual=h2)
                h = h3
                continue
            h1 = a.alloc(rows, c)
            rs1 = stat()
            r.conv(att, None, lin, bw.out1, h1, residual=h, rowstat_out=rs1)
            # cross-attention over the cached text K / V^T (shared by all images)
            h2 = a.alloc(rows, c)
            rs2 = stat()
            if xa is not None:
                # wide blocks: probabilities = softmax per head of LN(h1) G^T (one 128-column tile per head, softmax in the
                # GEMM epilogue), then h2 = P Z^T + bias + h1 -- two launches instead of three, fewer FLOPs for C >= 1024
                pr = a.alloc(rows, xa[0].n)
                r.conv(h1, None, lin, xa[0], pr, ln_part=rs1, act=L.ACT_SOFTMAX, softmax_cols=kt.shape[0])
                r.conv(pr, None, lin, xa[1], h2, residual=h1, rowstat_out=rs2)
            else:
                q = a.alloc(rows, c)
                r.conv(h1, None, lin, bw.q2, q, ln_part=rs1)
                r.attention(q, c, kt, c, vtt, vtt.shape[1], att, c, rows, kt.shape[0], heads, d, d ** -0.5)
                r.conv(att, None, lin, bw.out2, h2, residual=h1, rowstat_out=rs2)
            # GEGLU feed-forward
            f = a.alloc(rows, 4 * c)
            r.conv(h2, None, lin, bw.ff1, f, ln_part=rs2)
            h3 = a.alloc(rows, c)
            last = bi == len(tw.blocks) - 1
            rs = None if last else stat()  # the next block's norm1 statistics
            r.conv(f, None, lin, bw.ff2, h3, residual=h2, rowstat_out=rs)
            h = h3
        out = a.alloc(rows, c)
        r.conv(h, None, lin, tw.proj_out, out, residual=x, out2=out2, add2=add2,
               chanstat_out=stat_out)
        return out

    def _gn(self, r, x, x2, c0, c1, hw, groups, eps, gamma, beta, silu, out):
        # (round 1-2 could also take the statistics from the producing conv's epilogue (chanstat_out): slower than the separate,
        #  overlappable statistics kernel on MI355X, removed in round 3 -- chanstat_out remains for the reference-only AdaIN)
        r.groupnorm(x, x2, c0, c1, hw, groups, eps, gamma, beta, silu, out, batch=self.batch)

    def _vt_buffer(self, c, ldvt):
        # V^T buffers live outside the rewound arena: their key-padding columns must stay zero forever
        key = (c, ldvt, self._vt_count)
        self._vt_count += 1
        buf = self._vt_pool.get(key)
        if buf is None:
            buf = self.ops.zeros(c, ldvt)
            self._vt_pool[key] = buf
        return buf

    def _ref_site(self, r, ref: Optional[RefCtx], kind, i, j, c):
        """-> stat buffer the block's last conv must fill (or None) for the reference-only AdaIN at this block output"""
        if ref is None or not ref.gate(kind, i):
            return None
        st = self.arena.alloc(c, 2, dtype=torch.float32)
        if ref.mode == "write":
            ref.stats[(kind, i, j)] = st
        return st

    def _ref_adain(self, r, ref: Optional[RefCtx], kind, i, j, h, st, rows, c):
        """read pass: AdaIN the block output to the statistics the write pass banked for the same site"""
        if ref is None or st is None or ref.mode != "read" or (kind, i, j) not in ref.stats:
            return h
        o = self.arena.alloc(rows, c)
        r.adain(h, st, ref.stats[(kind, i, j)], rows, c, o)
        return o

    def _down_mid(self, r, net: NetWeights, step, h, sizes, ref: Optional[RefCtx] = None, temb=None):
        """conv_in output h -> (mid block output, skip list [(tensor, channels, level)])."""
        a = self.arena
        ch = net.cfg.block_out_channels
        skips = []
        for i, c in enumerate(ch):
            hh, ww = sizes[i]
            hw = hh * ww
            g3 = Geom.conv(hh, ww, batch=self.batch)
            for j, (rw, tw) in enumerate(net.down[i]):
                st = self._ref_site(r, ref, "down", i, j, c)
                h = self._resnet(r, rw, net, step, h, None, rw.cin, 0, hw, g3, temb=temb, stat_out=st if tw is None else None)
                if tw is not None:
                    h = self._transformer(r, tw, net, h, hw, ref=ref, stat_out=st)
                h = self._ref_adain(r, ref, "down", i, j, h, st, self.batch * hw, c)
                skips.append((h, c, i))
            ds = net.downsamplers[i]
            if ds is not None:
                h2, w2 = sizes[i + 1]
                o = a.alloc(self.batch * h2 * w2, c)
                r.conv(h, None, Geom.conv(hh, ww, stride=2, batch=self.batch), ds, o)
                h = o
                skips.append((o, c, i + 1))
        hh, ww = sizes[-1]
        hw = hh * ww
        g3 = Geom.conv(hh, ww, batch=self.batch)
        c = ch[-1]
        h = self._resnet(r, net.mid[0], net, step, h, None, c, 0, hw, g3, temb=temb)
        h = self._transformer(r, net.mid[1], net, h, hw, ref=ref)
        st = self._ref_site(r, ref, "mid", 0, 0, c)
        h = self._resnet(r, net.mid[2], net, step, h, None, c, 0, hw, g3, temb=temb, stat_out=st)
        h = self._ref_adain(r, ref, "mid", 0, 0, h, st, self.batch * hw, c)
        return h, skips

    def _unet_encoder(self, r, step, lat, sizes, ref: Optional[RefCtx] = None, temb=None):
        a, net = self.arena, self.unet
        ch = net.cfg.block_out_channels
        h0, w0 = sizes[0]
        x = a.alloc(self.batch * h0 * w0, ch[0])
        r.conv(lat, None, Geom.conv(h0, w0, batch=self.batch), net.conv_in, x)
        h, skips = self._down_mid(r, net, step, x, sizes, ref=ref, temb=temb)
        return h, [(x, ch[0], 0)] + skips

    def _unet_decoder(self, r, step, h, skips, sizes, eps_out, ref: Optional[RefCtx] = None, temb=None):
        a, net = self.arena, self.unet
        ch = net.cfg.block_out_channels
        h0, w0 = sizes[0]
        hw0 = h0 * w0
        skips = list(skips)
        nlev = len(ch)
        cprev = ch[-1]
        for i in range(nlev):
            lvl = nlev - 1 - i
            hh, ww = sizes[lvl]
            hw = hh * ww
            g3 = Geom.conv(hh, ww, batch=self.batch)
            for j, (rw, tw) in enumerate(net.up[i]):
                s, sc, slvl, *ev = skips.pop()
                assert slvl == lvl and rw.cin == cprev + sc, (slvl, lvl, rw.cin, cprev, sc)
                if ev and ev[0]:
                    r.wait(ev[0])  # this skip's ControlNet merge ran on the second stream
                st = self._ref_site(r, ref, "up", i, j, rw.cout)
                h = self._resnet(r, rw, net, step, h, s, cprev, sc, hw, g3, temb=temb, stat_out=st if tw is None else None)
                cprev = rw.cout
                if tw is not None:
                    h = self._transformer(r, tw, net, h, hw, ref=ref, stat_out=st)
                h = self._ref_adain(r, ref, "up", i, j, h, st, self.batch * hw, cprev)
            up = net.upsamplers[i]
            if up is not None:
                h2, w2 = sizes[lvl - 1]
                o = a.alloc(self.batch * h2 * w2, cprev)
                # nearest resize to the next skip's size folded into the conv's gather (Upsample2D)
                r.conv(h, None, Geom.conv(hh, ww, up_to=(h2, w2), batch=self.batch), up, o)
                h = o
        t = a.alloc(self.batch * hw0, ch[0])
        self._gn(r, h, None, ch[0], 0, hw0, net.cfg.groups, 1e-5, net.norm_out[0], net.norm_out[1], True, t)
        r.conv(t, None, Geom.conv(h0, w0, batch=self.batch), net.conv_out, eps_out, ldo=8)

    def _controlnet_encoder(self, r, step, lat, sizes, cond_emb):
        a, net = self.arena, self.cn
        ch = net.cfg.block_out_channels
        h0, w0 = sizes[0]
        x = a.alloc(self.batch * h0 * w0, ch[0])
        r.conv(lat, None, Geom.conv(h0, w0, batch=self.batch), net.conv_in, x, residual=cond_emb)
        h, skips = self._down_mid(r, net, step, x, sizes)
        return h, [(x, ch[0], 0)] + skips

    def _controlnet_merge(self, r, cn_mid, cn_skips, u_mid, u_skips, sizes, scale):
        """The 13 zero-convs, each writing UNet tensor + scale_i * (conv + bias) in one epilogue
        (ControlNetModel's guess-mode scaling, always on in the reference: lcm_controlnet.py:399,447, and the
        `down_block_additional_residuals` / `mid_block_additional_residual` adds of UNet2DConditionModel)."""
        a, net = self.arena, self.cn
        nres = len(cn_skips) + 1
        sc = self._cn_scale_consts  # fp32 [nres] in device memory: logspace(-1, 0, nres) * controlnet_scale
        assert sc.numel() >= nres
        # Only the mid-block merge is on the critical path; the decoder consumes the skips deepest first, one per ResnetBlock.
        # With the second stream free after the encoders' join, the 12 skip merges run there, in the order the decoder
        # needs them, each followed by a named event the consuming ResnetBlock waits for.
        side = self.overlap_controlnet and self.use_side_stream
        merged = [None] * len(cn_skips)
        # (a grouped launch exists for the buffer-load operand path only: every merge's channel count a multiple of 64 -- the shipped
        #  SD1.5 / SDXL / MINI configs; a ControlNet of other widths takes the one-by-one merges below: ADVICE r5)
        groupable = all(c % 64 == 0 for (_s, c, _l) in cn_skips) and net.cfg.block_out_channels[-1] % 64 == 0
        if not side and self.group_merges and groupable and hasattr(self.ops, "conv_group"):
            # The 13 merges do not depend on each other and are small (11.5 us of a lone frame each as launches of their own: leaving
            # the 12 skip merges out of a 4-step frame takes 0.55 ms off it): ONE grid per group of up to eight of them
            # (vsd_conv_gemm_group) -- first the mid block's and the deepest skips' (what the decoder needs first), then the rest.
            hh, ww = sizes[-1]
            rows = self.batch * hh * ww
            mid = a.alloc(rows, net.cfg.block_out_channels[-1])
            calls = [((cn_mid, None, Geom.linear(rows), net.zero_mid, mid), dict(out_scale_dev=sc[nres - 1:nres], residual=u_mid))]
            for i in reversed(range(len(cn_skips))):
                (s, c, lvl), (us, uc, ulvl) = cn_skips[i], u_skips[i]
                assert (c, lvl) == (uc, ulvl)
                hh, ww = sizes[lvl]
                rows = self.batch * hh * ww
                o = a.alloc(rows, c)
                calls.append(((s, None, Geom.linear(rows), net.zero_convs[i], o), dict(out_scale_dev=sc[i:i + 1], residual=us)))
                merged[i] = (o, c, lvl, None)
            gmax = 7
            for g0 in range(0, len(calls), gmax):
                r.conv_group(calls[g0:g0 + gmax])
            return mid, merged
        if side:
            r.signal("enc_done")
            r.use_stream(1)
            r.wait("enc_done")
        for i in reversed(range(len(cn_skips))):
            (s, c, lvl), (us, uc, ulvl) = cn_skips[i], u_skips[i]
            assert (c, lvl) == (uc, ulvl)
            hh, ww = sizes[lvl]
            rows = self.batch * hh * ww
            o = a.alloc(rows, c)
            r.conv(s, None, Geom.linear(rows), net.zero_convs[i], o, out_scale_dev=sc[i:i + 1], residual=us)
            ev = None
            if side:
                ev = f"skip{i}"
                r.signal(ev)
            merged[i] = (o, c, lvl, ev)
        if side:
            r.use_stream(0)
        hh, ww = sizes[-1]
        rows = self.batch * hh * ww
        c = net.cfg.block_out_channels[-1]
        mid = a.alloc(rows, c)
        r.conv(cn_mid, None, Geom.linear(rows), net.zero_mid, mid, out_scale_dev=sc[nres - 1:nres], residual=u_mid)
        return mid, merged

    @staticmethod
    def _zip_pairs(r, first, second):
        """two recorded call lists of one topology -> pairs; calls without a twin (another op, or one list longer) go out alone"""
        sync = Engine.SYNC_OPS
        first = [c for c in first if c[0].__name__ not in sync]
        second = [c for c in second if c[0].__name__ not in sync]
        for i in range(max(len(first), len(second))):
            a = first[i] if i < len(first) else None
            b = second[i] if i < len(second) else None
            if a is not None and b is not None and a[0].__name__ == b[0].__name__ and a[0].__name__ in ("conv", "groupnorm", "attention", "tail_a", "tail_b"):
                r.pair(a, b)
            else:
                for c in (a, b):
                    if c is not None:
                        r.calls.append(c)

    def _cond_embedding(self, r, ctrl, H, W):
        a, net = self.arena, self.cn
        h, hh, ww = ctrl, H, W
        for pw, stride in net.cond_convs:
            g = Geom.conv(hh, ww, stride=stride, batch=self.batch)
            o = a.alloc(g.m, pw.n)
            r.conv(h, None, g, pw, o, act=L.ACT_SILU)
            h, hh, ww = o, g.ho, g.wo
        g = Geom.conv(hh, ww, batch=self.batch)
        o = a.alloc(g.m, net.cond_out.n)
        r.conv(h, None, g, net.cond_out, o)
        return o

    def _taesd_block(self, r, blk, x, hh, ww):
        a = self.arena
        g = Geom.conv(hh, ww, batch=self.batch)
        c = blk[0].n
        t1 = a.alloc(g.m, c)
        r.conv(x, None, g, blk[0], t1, act=L.ACT_RELU)
        t2 = a.alloc(g.m, c)
        r.conv(t1, None, g, blk[1], t2, act=L.ACT_RELU)
        o = a.alloc(g.m, c)
        r.conv(t2, None, g, blk[2], o, residual=x, act=L.ACT_RELU | L.ACT_POST)
        return o

    def _encode(self, r, img8, H, W, out):
        a, v = self.arena, self.vae
        c = v.enc_in.n
        B = self.batch
        h = a.alloc(B * H * W, c)
        r.conv(img8, None, Geom.conv(H, W, batch=B), v.enc_in, h)
        for b in v.enc_blocks0:
            h = self._taesd_block(r, b, h, H, W)
        hh, ww = H, W
        for down, bs in v.enc_stages:
            g = Geom.conv(hh, ww, stride=2, batch=B)
            o = a.alloc(g.m, c)
            r.conv(h, None, g, down, o)
            h, hh, ww = o, g.ho, g.wo
            for b in bs:
                h = self._taesd_block(r, b, h, hh, ww)
        r.conv(h, None, Geom.conv(hh, ww, batch=B), v.enc_out, out, ldo=8)

    def _decode(self, r, z8, hh, ww, out):
        a, v = self.arena, self.vae
        c = v.dec_in.n
        B = self.batch
        h = a.alloc(B * hh * ww, c)
        r.conv(z8, None, Geom.conv(hh, ww, batch=B), v.dec_in, h, act=L.ACT_RELU)
        for bs, up in v.dec_stages:
            for b in bs:
                h = self._taesd_block(r, b, h, hh, ww)
            g = Geom.conv(hh, ww, up_to=(2 * hh, 2 * ww), batch=B)  # nn.Upsample(scale_factor=2) folded into the conv
            o = a.alloc(g.m, c)
            r.conv(h, None, g, up, o)
            h, hh, ww = o, g.ho, g.wo
        h = self._taesd_block(r, v.dec_last_block, h, hh, ww)
        r.conv(h, None, Geom.conv(hh, ww, batch=B), v.dec_out, out, ldo=8)

    # ---------------------------------------------------------------- prepare: build + capture
    def autotune(self, verbose: bool = False):
        """Pick (tile, split-K, reduction form) per distinct conv shape of the recorded program by timing the
        candidates on the GPU (the hipBLASLt/MIOpen "find" step, done on our own kernel)."""
        ops = self.ops
        if not hasattr(ops, "tune_conv"):
            return {}
        seen = {}
        # throughput-mode plans: candidates are timed with four lanes busy only offline / on request (ops.tune_lanes_online);
        # otherwise a shape without a throughput-mode entry takes its alone-timed choice, timed now if that is missing too
        mode = getattr(ops, "tune_mode", 0)
        online = mode == 0 or getattr(ops, "tune_lanes_online", False)
        try:
            # the two-stream form first: a twin pair of the one-stream form runs at the split its members have as launches of their own
            calls = self.program.calls + (self.program_serial.calls if self.program_serial is not self.program else [])
            for fn, a, k in calls:
                if fn.__name__ in ("conv_group", "pair") and hasattr(ops, "tune_group"):
                    ops.tune_mode = mode if online else 0  # (a group's form is timed alone either way; see ops.tune_group)
                    members, split = a[0], k.get("split")
                    if split == getattr(ops, "OWN_SPLIT", None):  # (members at their own splits: their own forms first)
                        for aa, kk in members:
                            mk = ops.conv_key_of(aa[2], aa[3], kk)
                            if mk not in ops.tile_override and kk.get("tile") is None:
                                seen[mk] = ops.tune_conv(aa, kk)[0]
                    if fn.__name__ == "pair":
                        if a[0][0].__name__ != "conv" or a[1][0].__name__ != "conv":
                            continue
                        if mode == 1:  # (throughput mode: a pair runs in its members' own form, ops.pair -- nothing to time)
                            continue
                        members = [(a[0][1], a[0][2]), (a[1][1], a[1][2])]
                        for aa, kk in members:  # (a pair the one-stream form alone holds: its members' own forms first)
                            mk = ops.conv_key_of(aa[2], aa[3], kk)
                            if mk not in ops.tile_override and kk.get("tile") is None:
                                seen[mk] = ops.tune_conv(aa, kk)[0]
                        split = ops.pair_split(*members[0], *members[1])
                        if split is None:
                            continue
                    key = ops.group_key(members, split)
                    if key not in seen and key not in ops.tile_override:
                        seen[key] = ops.tune_group(members, split=split)[0]
                        if verbose:
                            print("tune", key, "->", seen[key], flush=True)
                    continue
                if fn.__name__ != "conv":
                    continue
                ops.tune_mode = mode
                key = ops.conv_key_of(a[2], a[3], k)
                if key in seen or key in ops.tile_override or k.get("tile") is not None:
                    continue
                if not online:
                    ops.tune_mode = 0
                    key = ops.conv_key_of(a[2], a[3], k)
                    if key in seen or key in ops.tile_override:
                        continue
                best, table = ops.tune_conv(a, k)
                seen[key] = best
                if verbose:
                    print("tune", key, "->", best, flush=True)
        finally:
            if hasattr(ops, "tune_mode"):
                ops.tune_mode = mode
        return seen

    def prepare(self, H: int, W: int, steps: int, strength: float, controlnet_scale: float = 1.0,
                use_controlnet: bool = True, use_graph: Optional[bool] = None, autotune: bool = True, batch: int = 1,
                ref_mode: bool = False):
        """Fix the frame geometry and schedule; build the static program and capture it into a hipGraph
        (the reference's intent at videopipeline.py:35-47, `compile_model`).

        ref_mode: the reference-only variant (lcm_reference_pipeline.py:855-890; no ControlNet, one frame per launch): per
        step a WRITE pass of the UNet over the noised latents of the reference image (upload it into `ref_u8`) banks
        self-attention keys / values and block-output statistics, the READ pass over the frame's latents uses them.

        batch > 1: that many frames (of independent streams / sessions, or consecutive frames of one stream) go
        through every kernel together, stacked along the GEMM M dimension: one pass over the 2.45 GB of weights and
        one launch per layer serve all of them.  Each frame is still denoised independently (own GroupNorm statistics,
        own attention, same noise draws as a lone frame: the reference resets its RNG per frame)."""
        if batch < 1:
            raise ValueError("batch must be >= 1")
        self.batch = batch
        # The lock-step encoders are the one-stream form of EVERY program.  What a pair's grid looks like depends on the mode: a one-frame
        # program (latency mode) takes the pair's own table entry, timed alone; a coalesced launch on busy lanes (throughput mode) runs
        # the pair in the form its members have as launches of their own -- a latency-chosen form there occupies more workgroup-time
        # than the two launches (profiles/round5f_pairs_at_5x4.txt).  Measured on one box with / without pairs: 1 x 4 89.7 / 83.5,
        # 2 x 4 122.2 / 118.8, 3 x 4 133.8 / 131.3, 3 x 3 127.7 / 123.0, 4 x 4 138.8 / 138.3, 5 x 4 140.2 / 140.4 frames/s.
        self._twin_now = self.twin_encoders
        if hasattr(self.ops, "tune_mode"):
            self.ops.tune_mode = 1 if self.tune_for_lanes else 0
        if H % 8 or W % 8:
            raise ValueError("height and width must be multiples of 8 (TAESD / latent stride)")
        src = self._want if self._want is not None else self.family.get("prompt")
        if src is None:
            raise RuntimeError("set_text_embeds must be called before prepare")
        if self.pblock is None or self.pblock.layout is not src.layout:
            self.pblock = PromptBlock(self.ops, src.layout)
            self._installed = None
        if use_controlnet and self.cn is None:
            raise RuntimeError("no ControlNet weights loaded")
        if ref_mode and (batch != 1 or use_controlnet or (H // 8) * (W // 8) % 8):
            raise ValueError("ref_mode: one frame per launch, no ControlNet (the reference-only pipeline has none), "
                             "latent pixels a multiple of 8")
        if use_graph is not None:
            self.use_graph = use_graph
        ops = self.ops
        sched = LCMSchedule(strength, steps)
        n = len(sched)
        h0, w0 = H // 8, W // 8
        sizes = [(h0, w0)]
        for _ in range(len(self.ucfg.block_out_channels) - 1):
            ph, pw_ = sizes[-1]
            sizes.append(((ph + 1) // 2, (pw_ + 1) // 2))
        hw0 = h0 * w0
        self._destroy_graphs()
        self.arena = Arena(ops, chunk_bytes=max(256 << 20, _ru(batch * H * W * 64 * 2, 1 << 20)))  # >= one TAESD tensor
        self._vt_pool, self._vt_count = getattr(self, "_vt_pool", {}), 0
        # persistent per-frame I/O and constants
        B = batch
        frame_b = ops.zeros(B, H, W, 3, dtype=torch.uint8)
        out_b = ops.zeros(B, H, W, 3, dtype=torch.uint8)
        # public I/O buffers: [H][W][3] for a single frame (as before), [B][H][W][3] for a batch
        self.frame_u8 = frame_b[0] if B == 1 else frame_b
        self.out_u8 = out_b[0] if B == 1 else out_b
        self.edge_u8 = ops.zeros(B * H * W, dtype=torch.uint8)
        # Per-plan constants the captured graph reads from device memory: [0:2] add_noise coefficients, [2 + 6i : 8 + 6i]
        # the scheduler coefficients of step i, then the 13 ControlNet residual scales; plus the time-embedding projections
        # of every step.  `update_options` rewrites them in place -- a new strength / controlnet_scale needs no re-capture.
        ncn = 16
        if self.is_slot:  # computed by the parent engine's prepare / update_options
            c = self.shared.get("consts")
            assert c is not None and self.shared.get("n") == n, "prepare the parent engine with the same schedule first"
        else:
            c = self.shared.get("consts")
            if c is None or self.shared.get("n") != n:
                self.shared["consts"] = c = ops.zeros(2 + 6 * n + ncn, dtype=torch.float32)
                self.shared["n"] = n
            temb = self.shared.setdefault("temb", {})
            for name, net in [("unet", self.unet)] + ([("cn", self.cn)] if use_controlnet else []) + ([("ref", self.unet)] if ref_mode else []):
                if name not in temb or temb[name].shape[0] != n:
                    temb[name] = ops.zeros(n, net.temb_proj.n)
            self.shared["ref_mode"] = bool(ref_mode)
            self._write_constants(sched, controlnet_scale, use_controlnet)
        self._cn_scale_consts = c[2 + 6 * n:]
        # noise draws: the reference resets the global CPU generator to a fresh-Generator state on every
        # frame (videopipeline.py:126), so for a fixed shape the draws are the same every frame.
        draws, ref_draws = self.host_noise(n, h0, w0, ref=ref_mode)
        self.noise = ops.to_device(draws)
        a = self.arena
        enc_in = a.alloc(B * H * W, 8)
        x0 = a.alloc(B * hw0, 8)
        lat = [a.alloc(B * hw0, 8), a.alloc(B * hw0, 8)]
        eps = a.alloc(B * hw0, 8)
        den = a.alloc(B * hw0, 8)
        dec_in = a.alloc(B * hw0, 8)
        dec_out = a.alloc(B * H * W, 8)
        for t in (x0, lat[0], lat[1], eps, den, dec_in):
            ops.zero_(t)
        self.buffers = {"x0": x0, "lat": lat, "eps": eps, "denoised": den, "dec_in": dec_in, "dec_out": dec_out}
        r = Recorder(ops)
        img = lambda t, b, n: t[b * n:(b + 1) * n]  # noqa: E731  rows of image b
        r.preprocess_rgb(frame_b, B * H, W, enc_in)
        cond_emb = None
        self._ev_count = 0
        if use_controlnet:
            ctrl = a.alloc(B * H * W, 8)
            side0 = self.overlap_controlnet and self.use_side_stream
            if side0:  # Sobel + the conditioning embedding only feed the ControlNet encoder (which runs on stream 1 too)
                r.fork()
                r.use_stream(1)
            for b in range(B):  # the edge map is normalised by ITS frame's maximum (canny_gpu.py:39)
                r.sobel_control(frame_b[b], H, W, 0.11, 0.8, img(self.edge_u8, b, H * W), img(ctrl, b, H * W))  # videopipeline.py:109
            cond_emb = self._cond_embedding(r, ctrl, H, W)
            self.buffers["control"], self.buffers["cond_emb"] = ctrl, cond_emb
            if side0:
                r.use_stream(0)
        self._encode(r, enc_in, H, W, x0)
        if ref_mode:
            ref_b = ops.zeros(1, H, W, 3, dtype=torch.uint8)
            self.ref_u8 = ref_b[0]
            self.noise_ref = ops.to_device(ref_draws)
            ref_in, ref_x0, ref_xt, ref_eps = a.alloc(H * W, 8), a.alloc(hw0, 8), a.alloc(hw0, 8), a.alloc(hw0, 8)
            for t in (ref_x0, ref_xt, ref_eps):
                ops.zero_(t)
            r.preprocess_rgb(ref_b, H, W, ref_in)
            self._encode(r, ref_in, H, W, ref_x0)
            self.buffers["ref_x0"] = ref_x0
        # every frame gets the same draws: the reference resets its RNG per frame
        r.add_noise_dev(x0, self.noise[0], c[0:2], hw0, B, lat[0])
        mark = a.mark()
        for i in range(n):
            a.rewind(mark)
            self._vt_count = 0
            cur, nxt = lat[i & 1], lat[(i + 1) & 1]
            if use_controlnet:
                # the ControlNet encoder and the UNet encoder both depend only on the current latents: run them
                # on two streams (two parallel branches of the captured graph), join before the zero-convs
                # ... or in lock step on ONE stream: the ControlNet is a copy of the UNet encoder's topology, so the k-th call of
                # one is the k-th call of the other with another weight set -- each such pair goes out as one grid per kernel
                # (ops.pair): a quarter of a frame's launches less, and the small layers fill twice the chip.  Measured on MI355X
                # (one frame per launch, 512x512 4-step): a lone launch 20.8 ms on two streams | 21.3 ms in lock step; with three /
                # four lanes busy (no second stream to be had) 75.0 / 84.1 | 82.1 / 88.8 frames/s.  So both forms are recorded
                # -- same buffers, same results -- and `launch` takes the one its moment calls for.
                rc_, ru_ = Recorder(ops), Recorder(ops)
                cn_mid, cn_skips = self._controlnet_encoder(rc_, i, cur, sizes, cond_emb)
                u_mid, u_skips = self._unet_encoder(ru_, i, cur, sizes)
                two = Recorder(ops)
                if self.overlap_controlnet:
                    two.fork()
                    two.use_stream(1)
                two.calls += rc_.calls
                two.use_stream(0)
                two.calls += ru_.calls
                if self.overlap_controlnet:
                    two.join()
                if self._twin_now and hasattr(ops, "pair"):
                    twin = Recorder(ops)
                    self._zip_pairs(twin, ru_.calls, rc_.calls)
                    r.variants(two.calls, twin.calls)
                else:
                    r.variants(two.calls, two.calls)
                u_mid, u_skips = self._controlnet_merge(r, cn_mid, cn_skips, u_mid, u_skips, sizes, controlnet_scale)
            elif ref_mode:
                rc = RefCtx(self.ucfg)
                # ref_xt = add_noise(ref latents, fresh draw, t_i) (lcm_reference_pipeline.py:861-871); the coefficients of
                # timestep t_i are the first two of the step's scheduler coefficients
                r.add_noise_dev(ref_x0, self.noise_ref[i], c[2 + 6 * i:4 + 6 * i], hw0, 1, ref_xt)
                w_mid, w_skips = self._unet_encoder(r, i, ref_xt, sizes, ref=rc, temb=self.shared["temb"]["ref"])
                self._unet_decoder(r, i, w_mid, w_skips, sizes, ref_eps, ref=rc, temb=self.shared["temb"]["ref"])
                rc.mode = "read"
                u_mid, u_skips = self._unet_encoder(r, i, cur, sizes, ref=rc)
            else:
                u_mid, u_skips = self._unet_encoder(r, i, cur, sizes)
            self._unet_decoder(r, i, u_mid, u_skips, sizes, eps, ref=rc if ref_mode else None)
            nz = self.noise[i + 1] if sched.multistep else None
            last = i == n - 1
            r.lcm_step_dev(eps, cur, nz, c[2 + 6 * i:8 + 6 * i], hw0, B, nxt, den, dec_in if last else None)
        self._decode(r, dec_in, h0, w0, dec_out)
        r.postprocess_rgb(dec_out, 8, B * H * W, out_b)
        # program: what a lone launch runs (the ControlNet encoder on the side stream when `overlap_controlnet`);
        # program_serial: everything on the lane's own stream (the encoders in lock step when `twin_encoders`)
        self.program = r.flavor(0 if self.overlap_controlnet or not self._twin_now else 1)
        self.program_serial = r.flavor(1) if self._twin_now and self.overlap_controlnet and use_controlnet else self.program
        self.plan = dict(H=H, W=W, steps=steps, strength=strength, cn_scale=controlnet_scale, cn=use_controlnet, n=n, batch=B,
                         ref_mode=bool(ref_mode), tuned_for_lanes=bool(self.tune_for_lanes),
                         sizes=sizes, timesteps=sched.timesteps, n_ops=len(self.program.calls), arena_bytes=a.peak)
        # per-shape kernel configuration (timed once per shape, cached in ops.tile_override), warm-up, capture
        torch.cuda.synchronize() if torch.cuda.is_available() else None  # allocation fills vs. kernel streams
        self._sync_prompt()
        if autotune:  # (only shapes missing from the shared table are timed: a slot with the parent's batch size finds all)
            self.autotune()
        self.program.run()
        if self.program_serial is not self.program:
            self.program_serial.run()
        ops.synchronize()
        if self.use_graph:
            # two launch sequences of the SAME program (same kernels, same buffers): the ControlNet encoder on the lane's side
            # stream (a lone launch: ~4 ms less per frame), and everything on the lane's own stream (what a launch takes when
            # three or four lanes are busy: the side stream IS another lane's stream).  `launch(overlap=...)` picks per launch.
            self.graph = self._capture(self.program)
            self.plan["graphs"], self.plan["edges"] = ops.seq_count(self.graph)
            two_forms = self.plan["edges"] or self.program_serial is not self.program
            self.graph_serial = self._capture(self.program_serial, serial=True) if two_forms else self.graph
        return self.plan

    SYNC_OPS = ("use_stream", "fork", "join", "signal", "wait")

    @staticmethod
    def flat_calls(calls):
        """the recorded calls as single-op calls: the members of pairs and groups one by one (analysis scripts)"""
        for fn, a, k in calls:
            if fn.__name__ == "pair":
                yield a[0]
                yield a[1]
            elif fn.__name__ == "conv_group":
                for aa, kk in a[0]:
                    yield (fn.__self__.conv, aa, kk)
            else:
                yield (fn, a, k)

    def launches_by_kind(self, serial: bool = False):
        """kernel launches one replay of the recorded program (serial: of its one-stream form) issues, by kind ("pair_*": two twin calls in one grid;
        "convs_in_groups" counts members, not launches) -> (total, {kind: launches})"""
        ops, kinds = self.ops, {}

        def count(name, n=1):
            if n:
                kinds[name] = kinds.get(name, 0) + n

        def gn_launches(a, k=None):  # (csrc/norm.hip gn_try_fused: one launch for small images, else statistics + apply)
            if hasattr(ops, "groupnorm_launches"):  # (the library's own decision; the rule below serves the CPU op emulator of the tests)
                return ops.groupnorm_launches(a[2], a[3], a[4], a[5], (k or {}).get("batch", 1))
            c, hw, groups = a[2] + a[3], a[4], a[5]
            cpg = c // groups
            fused = ((hw <= 256 and cpg <= 40) or (hw <= 1024 and cpg <= 20)) and cpg in (40, 8, 16, 20, 4, 12, 10, 2, 6)
            return 1 if fused else 2

        def table(key):
            ent = ops.tile_override.get(key)
            if ent is None and key[-1] == 1:
                ent = ops.tile_override.get(key[:-1] + (0,))
            return ent

        def conv_reducer(a, k):
            ent = table(ops.conv_key_of(a[2], a[3], k))
            return int(ent is not None and ent[1] > 1 and not ent[2] and not a[3].tile128)

        for fn, a, k in (self.program_serial if serial else self.program).calls:
            name = fn.__name__
            if name in self.SYNC_OPS:
                continue
            if name == "pair":
                (fa, aa, ka), (fb, ab, kb) = a
                op = fa.__name__
                if op == "conv":
                    sp = ops.pair_split(aa, ka, ab, kb)
                    ent = table(ops.group_key([(aa, ka), (ab, kb)], sp)) if sp is not None else (ops.GROUP_ALONE,)
                    if sp is not None and getattr(ops, "tune_mode", 0) == 1:  # (throughput mode: the members' own form, ops.pair)
                        ent = ops._pair_default
                    if ent is None:
                        ent = (0, sp, True, 3)
                    if ent[0] != ops.GROUP_ALONE:
                        count("pair_conv")
                        count("pair_splitk_reduce", int(ent[1] > 1 and not ent[2]))
                    else:
                        count("conv", 2)
                        count("splitk_reduce", conv_reducer(aa, ka) + conv_reducer(ab, kb))
                elif op == "groupnorm":
                    count("pair_groupnorm")
                    count("pair_gn_second", gn_launches(aa, ka) - 1)
                else:
                    count("pair_" + op)
                continue
            if name == "conv_group":
                split = k.get("split")
                own = ops.own_splits(a[0]) if split == getattr(ops, "OWN_SPLIT", None) and hasattr(ops, "own_splits") else None
                ent = table(ops.group_key(a[0], split)) if hasattr(ops, "group_key") else None
                if (split is not None and own is None) or (ent is not None and ent[0] == ops.GROUP_ALONE):
                    for aa, kk in a[0]:  # (a group whose table entry -- or a member's own form -- sends the members out alone)
                        count("conv")
                        count("splitk_reduce", conv_reducer(aa, kk))
                    continue
                count(name)
                count("convs_in_groups", len(a[0]))
                if ent is not None and not ent[2] and ((own and max(own) > 1) or (not own and ent[1] > 1)):
                    count("group_splitk_reduce")
                continue
            count(name)
            if name == "conv":
                count("splitk_reduce", conv_reducer(a, k))
            elif name == "groupnorm":
                count("gn_second", gn_launches(a, k) - 1)
        return sum(v for k, v in kinds.items() if k != "convs_in_groups"), kinds

    def _capture(self, r: Recorder, serial: bool = False):
        """serial=True: every call on stream 0, no edges (the program's fork / join / signal / wait markers are dropped: in one
        in-order stream they hold by construction) -- one graph.
        The recorded program -> a launch sequence (include/vsd.h vsd_seq): every run of kernel calls on one stream becomes ONE
        single-branch hipGraph on that stream, every fork / join / signal / wait an event edge between the two streams, issued
        in program order by `vsd_seq_launch`.  A program without a second stream is one graph, as before.  (One graph with
        parallel branches is what rounds 1-3 captured; on this runtime two such graphs in flight serialise -- DESIGN.md
        section 3, "launches in flight".)"""
        ops = self.ops
        seq = ops.seq_create()
        cur, run, names = 0, [], {}

        def flush():
            nonlocal run
            if run:
                ops.seq_capture_begin(cur)
                try:
                    for fn, a, k in run:
                        fn(*a, **k)
                finally:
                    ops.seq_capture_end(seq, cur)
                run = []

        try:
            for fn, a, k in r.calls:
                name = getattr(fn, "__name__", "")
                if name not in self.SYNC_OPS:
                    run.append((fn, a, k))
                    continue
                if serial:
                    continue
                if name == "use_stream":
                    if a[0] != cur:
                        flush()
                        cur = a[0]
                    continue
                flush()
                if name == "fork":
                    ops.seq_wait(seq, 1, ops.seq_record(seq, 0))
                elif name == "join":
                    ops.seq_wait(seq, 0, ops.seq_record(seq, 1))
                elif name == "signal":
                    names[a[0]] = ops.seq_record(seq, cur)
                else:
                    ops.seq_wait(seq, cur, names[a[0]])
            flush()
        except Exception:
            ops.use_stream(0)
            ops.seq_destroy(seq)
            raise
        ops.use_stream(0)
        return seq

    def _destroy_graphs(self):
        g, gs = self.graph, getattr(self, "graph_serial", None)
        self.graph = self.graph_serial = None
        if g is not None:
            self.ops.seq_destroy(g)
        if gs is not None and gs is not g:
            self.ops.seq_destroy(gs)

    def _write_constants(self, sched: LCMSchedule, controlnet_scale: float, use_controlnet: bool):
        """Schedule- and option-dependent constants -> the device block / time-embedding tables the graphs read."""
        n = len(sched)
        vals = list(sched.add_noise_coef())
        for i in range(n):
            vals += [float(x) for x in sched.step_coef(i)]
        nres = 13 if self.cn is None else len(self.cn.zero_convs) + 1
        # ControlNetModel guess mode: logspace(-1, 0, 13) * conditioning_scale (always on: lcm_controlnet.py:399,447)
        vals += (torch.logspace(-1, 0, nres) * float(controlnet_scale)).tolist()
        c = self.shared["consts"]
        host = torch.zeros(c.numel(), dtype=torch.float32)
        host[:len(vals)] = torch.tensor(vals, dtype=torch.float32)
        self.ops.upload(c, host)
        for name, net in [("unet", self.unet)] + ([("cn", self.cn)] if (use_controlnet and self.cn is not None) else []):
            self._time_embeddings(net, sched, self.shared["temb"][name])
        if self.shared.get("ref_mode"):  # the reference-only WRITE pass: same timesteps, no guidance embedding
            self._time_embeddings(self.unet, sched, self.shared["temb"]["ref"], use_cond=False)
        self.ops.synchronize()

    def update_options(self, strength: float, controlnet_scale: float) -> bool:
        """New `strength` / `controlnet_scale` for the prepared plan WITHOUT re-capturing: the captured graph reads the
        scheduler coefficients, the ControlNet scales and the per-step time-embedding projections from device memory, so
        a slider drag (server.py:163-197: strength in steps of 0.02, controlnet_scale 0.05-3) costs a few small uploads and
        the ~10 tiny GEMMs of the time path.  Returns False when the new strength gives another NUMBER of timesteps (then
        the program itself changes: call `prepare`).  Call it on the parent engine, with no launch in flight; its slots
        follow (they share the constant block)."""
        if self.plan is None or self.is_slot:
            raise RuntimeError("update_options: prepare the parent engine first")
        sched = LCMSchedule(strength, self.plan["steps"])
        if len(sched) != self.plan["n"]:
            return False
        self._write_constants(sched, controlnet_scale, self.plan["cn"])
        self.plan.update(strength=strength, cn_scale=controlnet_scale, timesteps=sched.timesteps)
        return True

    @staticmethod
    def host_noise(n_steps: int, h: int, w: int, ref: bool = False):
        """The CPU draws of one frame, in the reference's order: draw 0 = prepare_latents noise
        (lcm_controlnet.py:331, generator not forwarded), then one torch.randn per scheduler step
        (:1033) when the schedule has more than one step.  fp32 [(n+1), 4, h*w].
        ref=True (reference-only mode): every step first draws the noise of the reference latents
        (lcm_reference_pipeline.py:861-863) from the same stream; returns (draws, ref_draws [n, 4, h*w])."""
        st = torch.get_rng_state()
        try:
            torch.manual_seed(0)
            torch.default_generator.set_state(torch.Generator(device="cpu").get_state())
            draws = [torch.randn(1, 4, h, w)]
            ref_draws = []
            for _ in range(n_steps):
                if ref:
                    ref_draws.append(torch.randn(1, 4, h, w))
                if n_steps > 1:
                    draws.append(torch.randn(1, 4, h, w))
            if n_steps <= 1:
                draws.append(torch.zeros(1, 4, h, w))
        finally:
            torch.set_rng_state(st)
        d = torch.cat(draws, dim=0).reshape(len(draws), 4, h * w).contiguous()
        if not ref:
            return d, None
        return d, torch.cat(ref_draws, dim=0).reshape(n_steps, 4, h * w).contiguous()

    # ---------------------------------------------------------------- per frame
    def launch(self, overlap: Optional[bool] = None):
        """Enqueue one frame's work (frame_u8 -> out_u8) on the lane's stream(s).  overlap: run the ControlNet encoder on the
        lane's side stream (default: `self.overlap_launch`); pass False when a launch of the lane that OWNS that stream
        (lane + 2 mod 4) may be in flight -- two busy queues on one command-processor pipe take turns (ops.HipOps)."""
        self._sync_prompt()
        if self.graph is not None:
            ov = self.overlap_launch if overlap is None else overlap
            self.ops.seq_launch(self.graph if ov else self.graph_serial)
        else:
            if hasattr(self.ops, "tune_mode"):
                self.ops.tune_mode = 1 if self.tune_for_lanes else 0
            ov = self.overlap_launch if overlap is None else overlap
            (self.program if ov else self.program_serial).run()

    def _want_shape(self):
        p = self.plan
        return (p["H"], p["W"], 3) if p["batch"] == 1 else (p["batch"], p["H"], p["W"], 3)

    def _staging(self):
        """Pinned host buffers of this plan's frame(s): H2D / D2H run as asynchronous DMA on the kernel stream instead of
        going through the runtime's pageable-memory staging copies."""
        key = tuple(self.frame_u8.shape)
        st = getattr(self, "_stage", None)
        if st is None or st[0] != key:
            pin = torch.cuda.is_available()
            st = (key, torch.empty(key, dtype=torch.uint8, pin_memory=pin), torch.empty(key, dtype=torch.uint8, pin_memory=pin),
                  torch.cuda.Event(enable_timing=True) if pin else None, torch.cuda.Event(enable_timing=True) if pin else None)
            self._stage = st
        return st

    def submit_u8(self, frame: np.ndarray, overlap: Optional[bool] = None):
        """Upload + enqueue one frame (or batch) without waiting: pair with `collect_u8`.  Lets the host prepare the
        next frames / post-process the previous ones while this one is on the GPU.  overlap: see `launch`."""
        want = self._want_shape()
        if frame.shape != want or frame.dtype != np.uint8:
            raise ValueError(f"frame must be uint8 {want}, got {frame.dtype} {frame.shape}")
        _, hin, _hout, e0, e1 = self._staging()
        hin.numpy()[...] = frame.reshape(hin.shape)
        self.ops.upload(self.frame_u8, hin)
        if e0 is not None:
            e0.record(self.ops.stream)
        self.launch(overlap)
        if e1 is not None:
            e1.record(self.ops.stream)

    def collect_u8(self) -> np.ndarray:
        """Wait for the frame(s) enqueued by the last `submit_u8` and bring them to the host (a fresh array)."""
        want = self._want_shape()
        _, _hin, hout, e0, e1 = self._staging()
        self.ops.download_into(hout, self.out_u8)
        if e0 is not None:
            self.last_gpu_ms = e0.elapsed_time(e1)  # the graph alone (between the H2D and the D2H copies)
        return hout.numpy().reshape(want).copy()

    def infer_u8(self, frame: np.ndarray) -> np.ndarray:
        """frame: uint8 [H][W][3] already cropped/resized by the caller -> uint8 [H][W][3]
        (prepared with batch B > 1: uint8 [B][H][W][3] -> [B][H][W][3])."""
        self.submit_u8(frame)
        return self.collect_u8()
