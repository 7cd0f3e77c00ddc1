"""Host side of the LCM schedule (product code; the device arithmetic is vsd_add_noise / vsd_lcm_step).

Mirrors LCMScheduler_X of the reference: betas/alphas_cumprod (/root/reference/diffusert/lcm/
lcm_controlnet.py:791-815), set_timesteps (:905-938), boundary-condition scalings (:940-946) and the
coefficients `step` (:948-1043) and `add_noise` (:1046-1071) use.  Everything here depends only on
(strength, steps), so it is computed once per option change and handed to the kernels as scalars.
"""
import math
from typing import List, Tuple

import numpy as np
import torch


def _alphas_cumprod(beta_start=0.00085, beta_end=0.012, n=1000) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


_ALPHAS_CUMPROD = None


def alphas_cumprod() -> torch.Tensor:
    global _ALPHAS_CUMPROD
    if _ALPHAS_CUMPROD is None:
        _ALPHAS_CUMPROD = _alphas_cumprod()
    return _ALPHAS_CUMPROD


def lcm_timesteps(strength: float, steps: int, origin_steps: int = 50, train_steps: int = 1000) -> List[int]:
    """LCMScheduler_X.set_timesteps.  May return fewer than `steps` entries (e.g. (0.05, 4) -> [39, 19]);
    an empty list (int(50*strength) == 0) is rejected because the reference would fail on it
    (lcm_controlnet.py:588, `denoised` undefined)."""
    c = train_steps // origin_steps
    origin = np.asarray(list(range(1, int(origin_steps * strength) + 1))) * c - 1
    skipping = max(len(origin) // steps, 1)
    ts = origin[::-skipping][:steps]
    if len(ts) == 0:
        raise ValueError(f"strength={strength} gives an empty LCM schedule")
    return [int(t) for t in ts]


def w_embedding(w: float, dim: int = 256) -> torch.Tensor:
    """Guidance-scale embedding (get_w_embedding, lcm_controlnet.py:347-368) -> fp32 [1, dim]."""
    wv = torch.tensor([w], dtype=torch.float32) * 1000.0
    half = dim // 2
    emb = torch.log(torch.tensor(10000.0)) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = wv[:, None] * emb[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)


def timestep_sinusoid(timesteps: List[int], dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): fp32 [len, dim] = [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = torch.tensor(timesteps, dtype=torch.float32)[:, None] * freqs[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


class LCMSchedule:
    def __init__(self, strength: float, steps: int):
        self.strength, self.steps = strength, steps
        self.timesteps = lcm_timesteps(strength, steps)
        self.ac = alphas_cumprod()

    def __len__(self):
        return len(self.timesteps)

    @property
    def multistep(self) -> bool:
        return len(self.timesteps) > 1

    def add_noise_coef(self) -> Tuple[float, float]:
        a = self.ac[self.timesteps[0]]
        return float(a ** 0.5), float((1 - a) ** 0.5)

    def step_coef(self, i: int) -> Tuple[float, float, float, float, float, float]:
        """(sqrt_a, sqrt_b, c_skip, c_out, sqrt_a_prev, sqrt_b_prev) for loop index i."""
        t = self.timesteps[i]
        tp = self.timesteps[i + 1] if i + 1 < len(self.timesteps) else t
        a_t, a_p = self.ac[t], self.ac[tp]
        tt = torch.tensor(t)
        sigma = 0.5
        c_skip = sigma ** 2 / ((tt / 0.1) ** 2 + sigma ** 2)
        c_out = (tt / 0.1) / ((tt / 0.1) ** 2 + sigma ** 2) ** 0.5
        return (float(a_t.sqrt()), float((1 - a_t).sqrt()), float(c_skip), float(c_out), float(a_p.sqrt()),
                float((1 - a_p).sqrt()))
