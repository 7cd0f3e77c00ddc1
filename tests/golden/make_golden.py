"""Generate golden known-answer vectors by EXECUTING the reference's own in-tree arithmetic.

Run only in the build container (needs /root/reference).  Output: tests/golden/lcm_scheduler.json.

The reference's hot-path arithmetic that lives in its own tree is
  * LCMScheduler_X  (diffusert/lcm/lcm_controlnet.py:713-1100): __init__, set_timesteps,
    get_scalings_for_boundary_condition_discrete, step, add_noise
  * LatentConsistencyModelPipeline_controlnet.get_w_embedding (lcm_controlnet.py:347-368)
  * the per-frame CPU RNG reset of VideoSDPipeline.infer (videopipeline.py:28-32,110-126)
`diffusers` is not installable here, so the module's *import statements* are satisfied with
empty placeholder modules (base classes / a register_to_config decorator that only records the
constructor arguments).  None of the placeholder code takes part in the arithmetic recorded
below: every number in the JSON is produced by the reference's functions running on torch/numpy.
No reference source is copied; only inputs and outputs are stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/diffusert"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lcm_scheduler.json")


class _AttrDict(dict):
    __getattr__ = dict.__getitem__


def _placeholder_modules():
    def mk(name):
        m = types.ModuleType(name)
        m.__path__ = []

        def _ga(attr, _n=name):
            if attr.startswith("__"):
                raise AttributeError(attr)
            return type(attr, (), {})

        m.__getattr__ = _ga
        sys.modules[name] = m
        return m

    names = [
        "diffusers", "diffusers.configuration_utils", "diffusers.image_processor", "diffusers.models",
        "diffusers.pipelines", "diffusers.pipelines.stable_diffusion",
        "diffusers.pipelines.stable_diffusion.safety_checker", "diffusers.utils",
        "diffusers.utils.torch_utils", "diffusers.pipelines.controlnet",
        "diffusers.pipelines.controlnet.multicontrolnet", "diffusers.schedulers",
        "diffusers.schedulers.scheduling_utils", "diffusers.pipelines.pipeline_utils", "diffusers.loaders",
    ]
    mods = {n: mk(n) for n in names}

    import functools
    import inspect

    def register_to_config(init):
        @functools.wraps(init)
        def wrapper(self, *a, **kw):
            sig = inspect.signature(init)
            ba = sig.bind(self, *a, **kw)
            ba.apply_defaults()
            cfg = {k: v for k, v in ba.arguments.items() if k != "self"}
            self.config = _AttrDict(cfg)
            init(self, *a, **kw)

        return wrapper

    mods["diffusers.configuration_utils"].register_to_config = register_to_config
    lg = types.SimpleNamespace(get_logger=lambda *_a, **_k: types.SimpleNamespace(
        warning=lambda *a, **k: None, info=lambda *a, **k: None))
    mods["diffusers"].logging = lg
    mods["diffusers.utils"].logging = lg
    return mods


def main():
    _placeholder_modules()
    sys.path.insert(0, REF)
    import lcm.lcm_controlnet as ref  # noqa: E402  (the reference module itself)

    sch = ref.LCMScheduler_X(beta_start=0.00085, beta_end=0.0120, beta_schedule="scaled_linear",
                             prediction_type="epsilon")
    g = {}
    # ---- timesteps table (strength, steps) -> list   [set_timesteps :905-938]
    cases = [(0.4, 20), (0.6, 4), (0.4, 4), (0.8, 4), (1.0, 4), (0.5, 4), (0.6, 1), (1.0, 1), (0.6, 8),
             (0.6, 12), (0.05, 4), (0.3, 2), (0.98, 12), (0.6, 3), (0.62, 4), (0.1, 1), (0.9, 7)]
    tt = []
    for s, n in cases:
        sch.set_timesteps(s, n, 50)
        tt.append({"strength": s, "steps": n, "timesteps": [int(x) for x in sch.timesteps.tolist()]})
    g["timesteps"] = tt
    # ---- alphas_cumprod (all 1000, as float32 hex for exactness)
    ac = sch.alphas_cumprod.numpy().astype(np.float32)
    g["alphas_cumprod_f32_hex"] = ac.tobytes().hex()
    g["alphas_cumprod_probe"] = {str(i): float(ac[i]) for i in (0, 19, 99, 199, 299, 399, 499, 599, 999)}
    # ---- boundary scalings
    sc = {}
    for t in (19, 99, 179, 319, 459, 599, 999):
        cs, co = sch.get_scalings_for_boundary_condition_discrete(torch.tensor(t))
        sc[str(t)] = [float(cs), float(co)]
    g["scalings"] = sc
    # ---- w embedding
    w = torch.tensor(7.5).repeat(1)
    e = ref.LatentConsistencyModelPipeline_controlnet.get_w_embedding(None, w, embedding_dim=256)
    g["w_embedding_7p5_f32_hex"] = e.numpy().astype(np.float32).tobytes().hex()
    # ---- RNG contract of VideoSDPipeline.infer (videopipeline.py:28-32, 126)
    fresh = torch.Generator(device="cpu")
    g["fresh_generator_initial_seed"] = int(fresh.initial_seed())
    init_state = fresh.get_state()
    rng = {}
    for seed in (23, 42):
        torch.manual_seed(seed).set_state(init_state)
        a = torch.randn(4)
        torch.manual_seed(seed).set_state(init_state)
        b = torch.randn(1, 4, 64, 64)
        c = torch.randn(1, 4, 64, 64)
        rng[str(seed)] = {"randn4": [float(x) for x in a], "draw0_sum": float(b.sum()),
                          "draw0_first4": [float(x) for x in b[0, 0, 0, :4]], "draw1_sum": float(c.sum())}
    g["rng"] = rng
    # ---- step / add_noise on seeded inputs, fp32 (CPU-oracle dtype) : 4-step schedule and 1-step schedule
    steps_out = []
    for (s, n) in [(0.6, 4), (0.6, 1), (0.05, 4)]:
        sch.set_timesteps(s, n, 50)
        ts = sch.timesteps
        gen = torch.Generator().manual_seed(1000 + n)
        sample0 = torch.randn(1, 4, 8, 8, generator=gen)
        noise0 = torch.randn(1, 4, 8, 8, generator=gen)
        noisy = sch.add_noise(sample0, noise0, ts[:1])
        lat = noisy
        rec = {"strength": s, "steps": n, "sample0": sample0.flatten().tolist(), "noise0": noise0.flatten().tolist(),
               "noisy": noisy.flatten().tolist(), "iters": []}
        torch.manual_seed(n).set_state(init_state)
        for i, t in enumerate(ts):
            eps = torch.randn(1, 4, 8, 8, generator=gen)
            prev, den = sch.step(eps, i, t, lat, return_dict=False)
            rec["iters"].append({"t": int(t), "eps": eps.flatten().tolist(), "prev": prev.flatten().tolist(),
                                 "denoised": den.flatten().tolist(), "prev_is_denoised": bool(prev is den)})
            lat = prev
        steps_out.append(rec)
    g["step_traces"] = steps_out
    # ---- dtype behaviour (fp16 in)
    sch.set_timesteps(0.6, 4, 50)
    x16 = torch.zeros(1, 4, 2, 2, dtype=torch.float16)
    prev, den = sch.step(x16, 0, sch.timesteps[0], x16, return_dict=False)
    g["dtype_fp16_in"] = {"prev": str(prev.dtype), "denoised": str(den.dtype)}
    with open(OUT, "w") as f:
        json.dump(g, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
