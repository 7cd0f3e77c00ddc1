"""LCMScheduler_X restatement (reference: diffusert/lcm/lcm_controlnet.py:766-832, 905-946, 948-1071).

fp32 torch tensors on CPU, same operation order as the reference so that results agree to the ulp
with the golden traces in tests/golden/lcm_scheduler.json.  TEST INFRASTRUCTURE (see oracle/__init__).
"""
import numpy as np
import torch


class LCMSchedulerOracle:
    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000):
        # lcm_controlnet.py:791-801 ("scaled_linear") and :814-815
        self.num_train_timesteps = num_train_timesteps
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, strength, num_inference_steps, lcm_origin_steps=50):
        # lcm_controlnet.py:929-938
        c = self.num_train_timesteps // lcm_origin_steps
        origin = np.asarray(list(range(1, int(lcm_origin_steps * strength) + 1))) * c - 1
        skipping = max(len(origin) // num_inference_steps, 1)
        ts = origin[::-skipping][:num_inference_steps]
        self.timesteps = torch.from_numpy(ts.copy().astype(np.int64))
        return self.timesteps

    @staticmethod
    def scalings(t):
        # lcm_controlnet.py:940-946
        sigma_data = 0.5
        c_skip = sigma_data ** 2 / ((t / 0.1) ** 2 + sigma_data ** 2)
        c_out = (t / 0.1) / ((t / 0.1) ** 2 + sigma_data ** 2) ** 0.5
        return c_skip, c_out

    def add_noise(self, original, noise, timesteps):
        # lcm_controlnet.py:1046-1071
        ac = self.alphas_cumprod.to(dtype=original.dtype)
        sa = (ac[timesteps] ** 0.5).flatten()
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.dim() < original.dim():
            sa = sa.unsqueeze(-1)
            sb = sb.unsqueeze(-1)
        return sa * original + sb * noise

    def step(self, model_output, timeindex, timestep, sample):
        """Returns (prev_sample, denoised).  Draws torch.randn from the GLOBAL CPU generator when the
        schedule has more than one step (lcm_controlnet.py:1032-1036), including on the last step."""
        prev_i = timeindex + 1
        prev_t = self.timesteps[prev_i] if prev_i < len(self.timesteps) else timestep
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        c_skip, c_out = self.scalings(timestep)
        pred_x0 = (sample - b_t.sqrt() * model_output) / a_t.sqrt()
        denoised = c_out * pred_x0 + c_skip * sample
        if len(self.timesteps) > 1:
            noise = torch.randn(model_output.shape)
            prev = a_prev.sqrt() * denoised + b_prev.sqrt() * noise
        else:
            prev = denoised
        return prev, denoised


def w_embedding(w, embedding_dim=256, dtype=torch.float32):
    """get_w_embedding (lcm_controlnet.py:347-368); w: 1-D tensor."""
    w = w * 1000.0
    half = embedding_dim // 2
    emb = torch.log(torch.tensor(10000.0)) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=dtype) * -emb)
    emb = w.to(dtype)[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb
