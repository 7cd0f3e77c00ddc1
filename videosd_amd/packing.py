"""Weight re-layout for the HIP kernels (device-agnostic torch ops; runs once at load time).

Conv / linear weights become fp16 [N][Kp] matrices with k = (ky, kx, c) and Kp = K rounded up to 64
(zero filled), which is the B operand layout of vsd_conv_gemm (include/vsd.h).
"""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class PackedConv:
    weight: torch.Tensor            # fp16 [n][kp]
    bias: Optional[torch.Tensor]    # fp16 [n] or None
    n: int
    k: int
    kp: int
    cin: int                        # (padded) input channels per tap
    ksize: int
    geglu: bool = False
    ln_s: Optional[torch.Tensor] = None   # fp32 [n]: sum_k (W*gamma)[n][k]   (fused input LayerNorm, include/vsd.h)
    ln_t: Optional[torch.Tensor] = None   # fp32 [n]: sum_k beta[k] W[n][k] + bias[n]
    tile128: bool = False                 # the epilogue needs whole 128-column tiles and no split-K (tile softmax)
    weight_frag: Optional[torch.Tensor] = None  # the same weights fragment-major (pack_mfma_frag), for csrc/fused_tail.hip

    @property
    def n_out(self) -> int:
        return self.n // 2 if self.geglu else self.n


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _finish(w2d: torch.Tensor, bias, cin: int, ksize: int, geglu=False) -> PackedConv:
    n, k = w2d.shape
    kp = _round_up(k, 64)
    if kp != k:
        w2d = torch.nn.functional.pad(w2d, (0, kp - k))
    return PackedConv(w2d.to(torch.float16).contiguous(), None if bias is None else bias.to(torch.float16).contiguous(),
                      n, k, kp, cin, ksize, geglu)


def pack_conv(weight: torch.Tensor, bias=None, cin_pad: Optional[int] = None) -> PackedConv:
    """diffusers conv weight [Cout, Cin, kh, kw] -> [Cout][(ky,kx,c)]; Cin optionally zero-padded."""
    cout, cin, kh, kw = weight.shape
    assert kh == kw
    w = weight.permute(0, 2, 3, 1)  # [Cout, kh, kw, Cin]
    if cin_pad is not None and cin_pad != cin:
        w = torch.nn.functional.pad(w, (0, cin_pad - cin))
        cin = cin_pad
    assert cin % 8 == 0, "input channels must be a multiple of 8 (pad with cin_pad)"
    return _finish(w.reshape(cout, kh * kw * cin), bias, cin, kh)


def pack_linear(weight: torch.Tensor, bias=None) -> PackedConv:
    assert weight.shape[1] % 8 == 0
    return _finish(weight, bias, weight.shape[1], 1)


def pack_linear_cat(weights, biases=None) -> PackedConv:
    """Row-concatenate several linears sharing the same input (fused QKV / KV / per-resblock time projections)."""
    w = torch.cat(list(weights), dim=0)
    b = None if biases is None else torch.cat(list(biases), dim=0)
    return pack_linear(w, b)


def pack_geglu(weight: torch.Tensor, bias: torch.Tensor) -> PackedConv:
    """ff.net.0.proj [8C, C]: rows [0,4C) = hidden, [4C,8C) = gate.  Tile-pack so that every 128-row tile
    holds 64 hidden rows followed by their 64 gate rows (VSD_ACT_GEGLU epilogue)."""
    n, k = weight.shape
    f = n // 2
    assert f % 64 == 0
    wh = weight[:f].reshape(f // 64, 64, k)
    wg = weight[f:].reshape(f // 64, 64, k)
    w = torch.stack([wh, wg], dim=1).reshape(n, k)
    b = torch.stack([bias[:f].reshape(f // 64, 64), bias[f:].reshape(f // 64, 64)], dim=1).reshape(n)
    return _finish(w, b, k, 1, geglu=True)


def _fold_ln(weight: torch.Tensor, bias, gamma: torch.Tensor, beta: torch.Tensor):
    """LN(x) W^T + b == rstd * (x (W*gamma)^T - mean * s) + t  with s, t as below.  s is summed over the SAME
    fp16-rounded values the GEMM multiplies with, so that a constant row (x = mean) cancels exactly."""
    w32, g, b = weight.float(), gamma.float(), beta.float()
    wp = (w32 * g[None, :]).to(torch.float16)
    s = wp.float().sum(dim=1)
    t = (w32 * b[None, :]).sum(dim=1)
    if bias is not None:
        t = t + bias.float()
    return wp, s, t


def pack_linear_ln(weights, biases, gamma, beta) -> PackedConv:
    """Row-concatenated linears that consume LayerNorm(x): the norm is folded into weights + epilogue vectors."""
    w = torch.cat(list(weights), dim=0)
    b = None
    if biases is not None:
        b = torch.cat([bi if bi is not None else torch.zeros(wi.shape[0], dtype=wi.dtype, device=wi.device)
                       for wi, bi in zip(weights, biases)], dim=0)
    wp, s, t = _fold_ln(w, b, gamma, beta)
    p = _finish(wp, None, w.shape[1], 1)
    p.ln_s, p.ln_t = s.contiguous(), t.contiguous()
    return p


def pack_geglu_ln(weight: torch.Tensor, bias: torch.Tensor, gamma, beta) -> PackedConv:
    n, k = weight.shape
    f = n // 2
    assert f % 64 == 0
    wp, s, t = _fold_ln(weight, bias, gamma, beta)

    def tile(x):  # [n, ...] -> 64 hidden rows then their 64 gate rows per 128-row tile
        tail = x.shape[1:]
        h = x[:f].reshape(f // 64, 64, *tail)
        g = x[f:].reshape(f // 64, 64, *tail)
        return torch.stack([h, g], dim=1).reshape(n, *tail)

    p = _finish(tile(wp), None, k, 1, geglu=True)
    p.ln_s, p.ln_t = tile(s).contiguous(), tile(t).contiguous()
    return p


def pack_cross_attention(k: torch.Tensor, v: torch.Tensor, wq: torch.Tensor, wo: torch.Tensor, bo, gamma, beta, heads: int,
                         group: int = 128):
    """Cross-attention over a FIXED key set (the text tokens) as two plain GEMMs ("absorbed" form).

        out = Concat_h[ softmax(scale * q_h K_h^T) V_h ] Wo^T + bo,   q = LN(x) Wq^T
            = softmax_h( LN(x) G^T ) Z^T + bo
        G[h*group + j, :] = scale * sum_d K_h[j, d] Wq[h*dh + d, :]      (j < tl, zero rows above)
        Z[:, h*group + j] = sum_d Wo[:, h*dh + d] V_h[j, d]

    k, v: [tl, C] key / value projections of the text (Attention.to_k / to_v of attn2), wq / wo: attn2.to_q / to_out.0,
    gamma / beta: the LayerNorm in front (norm2), folded into G like every LN-consuming layer (pack_linear_ln).  Returns
    (PackedConv G' with ln_s / ln_t: N = heads*group, K = C;  PackedConv Z with bias: N = C, K = heads*group).  The first
    GEMM runs with the VSD_ACT_SOFTMAX epilogue (softmax_cols = tl), the second is an ordinary linear layer with the
    residual in its epilogue.  One head per 128-column tile: C >= heads*group/2 keeps the FLOP count at or below the
    three-kernel form (q projection, attention, out projection); the engine uses it for C >= 640."""
    tl, c = k.shape
    dh = c // heads
    assert tl <= group and c % heads == 0
    kf, vf, wqf, wof = k.float(), v.float(), wq.float(), wo.float()
    scale = dh ** -0.5
    g = torch.zeros(heads * group, c, dtype=torch.float32)
    z = torch.zeros(c, heads * group, dtype=torch.float32)
    for h in range(heads):
        sl = slice(h * dh, (h + 1) * dh)
        g[h * group:h * group + tl] = scale * (kf[:, sl] @ wqf[sl, :])
        z[:, h * group:h * group + tl] = wof[:, sl] @ vf[:, sl].t()
    xa1 = pack_linear_ln([g], None, gamma, beta)
    xa1.tile128 = True
    xa2 = pack_linear(z, bo)
    return xa1, xa2


def pack_mfma_frag(w2d: torch.Tensor) -> torch.Tensor:
    """[N][K] fp16 (N % 16 == 0, K % 32 == 0) -> fragment-major: blocks [N/16][K/32] of [4 (k/8)][16 (n)][8] halfs, so that
    lane l = n + 16 q of a 16x16x32 MFMA B fragment reads 16 contiguous bytes at offset 16 l of its 1 KB block
    (csrc/fused_tail.hip load_b: one wave-instruction = one contiguous KB of weights)."""
    n, k = w2d.shape
    assert n % 16 == 0 and k % 32 == 0
    v = w2d.reshape(n // 16, 16, k // 32, 4, 8)          # [nb][n][kb][q][e]
    return v.permute(0, 2, 3, 1, 4).contiguous().reshape(-1)  # [nb][kb][q][n][e]


def add_frag(p: PackedConv) -> PackedConv:
    p.weight_frag = pack_mfma_frag(p.weight[:, :p.k].to(torch.float16))
    return p
