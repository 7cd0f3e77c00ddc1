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
