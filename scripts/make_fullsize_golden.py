"""tests/golden/fullsize_oracle.npz: the CPU oracle's output for the slowest full-size parity cases (1-3 minutes of fp32
oracle each on the GPU box's 128 cores), so that the GPU suite checks them in seconds.

    gpurun -- 'python scripts/make_fullsize_golden.py gpurun_out/fullsize_oracle.npz'   then copy to tests/golden/

Runs where the tests run (the seeded weights are synthesised with DEVICE generators: weights.synthesize(device="cuda")),
with exactly the inputs of tests/test_sdxl_gpu.py::test_sdxl_1024_four_step_matches_oracle,
tests/test_pipeline_gpu.py::test_baseline_config5_768_eight_step_scale2_matches_oracle and
...::test_reference_only_mode_512_four_step_matches_oracle.  Stored per case: the final denoised
latents and the TAESD-encoded input latents (fp16) and every second row / column of the output image (uint8) -- the tests
compute their latent rel-L2, mean |diff| and PSNR against these.  VSD_LIVE_ORACLE=1 makes the tests run the oracle instead.

The fixture carries `guard_sha256` / `guard_files`: the digest of the oracle sources, videosd_amd/weights.py, videosd_amd/config.py and
the case parameters it was computed from (tests/golden_guard.py); tests/test_oracle_golden.py fails when they no longer match.

    python scripts/make_fullsize_golden.py OUT --only mini64     the small CPU-weights case alone (runs anywhere, seconds),
                                                                 merged into the arrays OUT already holds
    python scripts/make_fullsize_golden.py OUT --stamp           only re-stamp OUT with the current digest (when a guarded file
                                                                 changed in a way that cannot change the oracle's numbers)"""
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pipeline import OraclePipeline  # noqa: E402
from videosd_amd import config as C, weights as W  # noqa: E402
import json  # noqa: E402

import golden_guard as G  # noqa: E402
from test_pipeline_gpu import _cpu, _frame  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out = args[0] if args else G.GOLDEN_FULLSIZE
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
res = {}
if (only or "--stamp" in sys.argv) and os.path.exists(out):
    with np.load(out) as z:
        res = {k: z[k] for k in z.files}


def save():
    total, per = G.digests()
    res["guard_sha256"] = np.array(total)
    res["guard_files"] = np.array(json.dumps(per, sort_keys=True))
    np.savez_compressed(out, **res)
    print(out, os.path.getsize(out), "bytes, guard", total[:16])


if "--stamp" in sys.argv:
    save()
    sys.exit(0)

# the small case on CPU-generator weights (tests/golden_guard.py CASES["mini64"])
c = G.CASES["mini64"]
wu, wc = W.synthesize(W.unet_spec(C.MINI_UNET), "unet."), W.synthesize(W.controlnet_spec(C.MINI_CONTROLNET), "cn.")
text = (torch.randn(77, C.MINI_UNET.cross_dim, generator=torch.Generator().manual_seed(c["text_seed"])) * 0.5).half()
orc = OraclePipeline(C.MINI_UNET, C.MINI_CONTROLNET, wu, wc, W.synthesize(W.taesd_spec(C.TAESD), "vae."))
img = np.asarray(orc.infer(Image.fromarray(_frame(c["H"], c["W"], seed=c["frame_seed"]), "RGB"), text[None].float(), height=c["H"],
                           width=c["W"], strength=c["strength"], steps=c["steps"], seed=23, controlnet_scale=c["cn_scale"],
                           use_controlnet=True, keep_trace=True))
res["mini64_image_half"] = img[::2, ::2].copy()
res["mini64_denoised"] = orc.trace["denoised"][-1][0].half().numpy()
res["mini64_init_latents"] = orc.trace["init_latents"][0].half().numpy()
if only == "mini64":
    save()
    sys.exit(0)
del orc, wu, wc


def stress512(case="stress512"):
    """BASELINE configs[1] on the range-stress weight sets (tests/golden_guard.py CASES["stress512"] / ["stress512m"])"""
    c = G.CASES[case]
    t0 = time.time()
    wu = _cpu(W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda", stress=c["stress"]))
    wc = _cpu(W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda", stress=c["stress"]))
    wv = _cpu(W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda"))
    text = (torch.randn(77, C.SD15_UNET.cross_dim, generator=torch.Generator().manual_seed(c["text_seed"])) * 0.5).half()
    orc = OraclePipeline(C.SD15_UNET, C.SD15_CONTROLNET, wu, wc, wv)
    img = np.asarray(orc.infer(Image.fromarray(_frame(c["H"], c["W"], seed=c["frame_seed"]), "RGB"), text[None].float(), height=c["H"],
                               width=c["W"], strength=c["strength"], steps=c["steps"], seed=23, controlnet_scale=c["cn_scale"],
                               use_controlnet=True, keep_trace=True))
    res[case + "_image_half"] = img[::2, ::2].copy()
    res[case + "_denoised"] = orc.trace["denoised"][-1][0].half().numpy()
    res[case + "_init_latents"] = orc.trace["init_latents"][0].half().numpy()
    den = orc.trace["denoised"][-1][0]
    # ... and the same frame with every layer output rounded to fp16 (oracle.nets.EMULATE_FP16): how far fp16 STORAGE alone moves
    # this network's output on these weights -- the yardstick the HIP path is held against on this set
    img16 = np.asarray(orc.infer(Image.fromarray(_frame(c["H"], c["W"], seed=c["frame_seed"]), "RGB"), text[None].float(), height=c["H"],
                                 width=c["W"], strength=c["strength"], steps=c["steps"], seed=23, controlnet_scale=c["cn_scale"],
                                 use_controlnet=True, keep_trace=True, emulate_fp16=True))
    den16 = orc.trace["denoised"][-1][0]
    res[case + "_fp16emu_denoised"] = den16.half().numpy()
    res[case + "_fp16emu_image_half"] = img16[::2, ::2].copy()
    d16 = np.abs(img16.astype(int) - img.astype(int))
    print(f"{case} fp16-storage emulation vs fp32: denoised rel-L2 {float((den16 - den).norm() / den.norm()):.3e}, image mean |diff| {d16.mean():.2f} LSB", flush=True)
    print(f"{case}: {time.time() - t0:.0f} s; denoised latents mean {float(den.mean()):.3f} std {float(den.std()):.3f} max |x| {float(den.abs().max()):.2f}; "
          f"image mean {img.mean():.1f} std {img.std():.1f}", flush=True)


if only in ("stress512", "stress512m", "stress"):
    for case in (("stress512", "stress512m") if only == "stress" else (only,)):
        stress512(case)
    save()
    sys.exit(0)

wv = _cpu(W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda"))

t0 = time.time()
wu = _cpu(W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda"))
wc = _cpu(W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda"))
text = (torch.randn(77, C.SD15_UNET.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5).half()
orc = OraclePipeline(C.SD15_UNET, C.SD15_CONTROLNET, wu, wc, wv)
H = W_ = 768
img = np.asarray(orc.infer(Image.fromarray(_frame(H, W_, seed=41), "RGB"), text[None].float(), height=H, width=W_, strength=0.6,
                           steps=8, seed=23, controlnet_scale=2.0, use_controlnet=True, keep_trace=True))
res["config5_image_half"] = img[::2, ::2].copy()
res["config5_denoised"] = orc.trace["denoised"][-1][0].half().numpy()
res["config5_init_latents"] = orc.trace["init_latents"][0].half().numpy()
print(f"config5 768x768 8-step scale 2: {time.time() - t0:.0f} s", flush=True)
# the reference-only mode at 512x512, 4 steps (tests/test_pipeline_gpu.py::test_reference_only_mode_512_four_step_matches_oracle)
t0 = time.time()
H = W_ = 512
img = np.asarray(orc.infer(Image.fromarray(_frame(H, W_, seed=51), "RGB"), text[None].float(), height=H, width=W_, strength=0.6,
                           steps=4, seed=23, ref_image=Image.fromarray(_frame(H, W_, seed=52), "RGB"), keep_trace=True))
res["ref512_image_half"] = img[::2, ::2].copy()
res["ref512_denoised"] = orc.trace["denoised"][-1][0].half().numpy()
print(f"reference-only 512x512 4-step: {time.time() - t0:.0f} s", flush=True)
del orc, wu, wc

t0 = time.time()
cfg = C.SDXL_UNET
wx = _cpu(W.synthesize(W.unet_spec(cfg), "sdxl.", device="cuda"))
g = torch.Generator().manual_seed(11)
text = (torch.randn(77, cfg.cross_dim, generator=g) * 0.5).half()
pooled = (torch.randn(cfg.add_pooled_dim, generator=g) * 0.5).half()
orc = OraclePipeline(cfg, None, wx, None, wv)
H = W_ = 1024
img = np.asarray(orc.infer(Image.fromarray(_frame(H, W_, seed=2), "RGB"), text[None].float(), height=H, width=W_, strength=0.6,
                           steps=4, seed=23, use_controlnet=False, keep_trace=True, pooled=pooled))
res["sdxl1024_image_half"] = img[::2, ::2].copy()
res["sdxl1024_denoised"] = orc.trace["denoised"][-1][0].half().numpy()
print(f"SDXL 1024x1024 4-step: {time.time() - t0:.0f} s", flush=True)
stress512("stress512")
stress512("stress512m")
save()
