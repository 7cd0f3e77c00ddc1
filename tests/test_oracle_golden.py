"""The oracle's scheduler / w-embedding / RNG contract against the golden vectors produced by the
reference's own code (tests/golden/make_golden.py).  Exact for integers, <= 1 ulp for fp32."""
import json
import os

import numpy as np
import torch

from oracle.scheduler import LCMSchedulerOracle, w_embedding
from oracle.pipeline import reset_rng

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lcm_scheduler.json")))


def _ulp_close(a, b, ulps=1):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b))))


def test_timesteps_table():
    s = LCMSchedulerOracle()
    for c in G["timesteps"]:
        assert s.set_timesteps(c["strength"], c["steps"]).tolist() == c["timesteps"], c


def test_alphas_cumprod_bit_exact():
    s = LCMSchedulerOracle()
    ref = np.frombuffer(bytes.fromhex(G["alphas_cumprod_f32_hex"]), dtype=np.float32)
    assert np.array_equal(s.alphas_cumprod.numpy(), ref)


def test_scalings():
    for t, (cs, co) in G["scalings"].items():
        a, b = LCMSchedulerOracle.scalings(torch.tensor(int(t)))
        assert _ulp_close(float(a), cs) and _ulp_close(float(b), co)


def test_w_embedding():
    ref = np.frombuffer(bytes.fromhex(G["w_embedding_7p5_f32_hex"]), dtype=np.float32).reshape(1, 256)
    e = w_embedding(torch.tensor(7.5).repeat(1), 256).numpy()
    assert np.array_equal(e, ref)


def test_rng_contract():
    assert torch.Generator(device="cpu").initial_seed() == G["fresh_generator_initial_seed"]
    for seed, r in G["rng"].items():
        reset_rng(int(seed))
        assert torch.randn(4).tolist() == r["randn4"]
        reset_rng(int(seed))
        b = torch.randn(1, 4, 64, 64)
        c = torch.randn(1, 4, 64, 64)
        assert float(b.sum()) == r["draw0_sum"] and b[0, 0, 0, :4].tolist() == r["draw0_first4"]
        assert float(c.sum()) == r["draw1_sum"]


def test_step_traces():
    for rec in G["step_traces"]:
        s = LCMSchedulerOracle()
        ts = s.set_timesteps(rec["strength"], rec["steps"])
        x0 = torch.tensor(rec["sample0"]).view(1, 4, 8, 8)
        n0 = torch.tensor(rec["noise0"]).view(1, 4, 8, 8)
        lat = s.add_noise(x0, n0, ts[:1])
        assert _ulp_close(lat.flatten().numpy(), rec["noisy"])
        reset_rng(rec["steps"])
        for i, (t, it) in enumerate(zip(ts, rec["iters"])):
            assert int(t) == it["t"]
            eps = torch.tensor(it["eps"]).view(1, 4, 8, 8)
            prev, den = s.step(eps, i, t, lat)
            assert _ulp_close(den.flatten().numpy(), it["denoised"])
            assert _ulp_close(prev.flatten().numpy(), it["prev"], ulps=2)
            assert (prev is den) == it["prev_is_denoised"]
            lat = prev
