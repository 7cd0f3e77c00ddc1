"""The full-size CPU oracle runs of the GPU suite, started AHEAD of the tests that check against them.

The four full-size parity tests (BASELINE configs 2, 4 and SDXL 1024x1024, plus the reference-only mode at 512x512)
spend 1-3 minutes each inside the fp32 CPU oracle while the GPU idles.  `start_all()` (tests/conftest.py, at the start of
a `-m gpu` session on a GPU box) runs those oracle calls in child processes -- own RNG state each: the oracle draws
from the global CPU generator like the reference does -- while the other GPU tests run; the consumer test then picks
the result up (`result(name)`), or computes it inline with the very same function (`JOBS[name]`) when no child was
started (a single test run by hand) or the child failed.  Test infrastructure only: nothing here is on the product path.

    python tests/oracle_ahead.py <job> <out.npz>      (what the child processes run)
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def frame(h, w, seed=1):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = ((xx * 5 + yy * 3) % 256).astype(np.uint8)[..., None]
    return (base // 2 + grad // 2).astype(np.uint8)


# ---------------------------------------------------------------- the oracle calls (shared by child and inline path)
def sd15_infer(orc, text, H, W, steps, cn, cn_scale, frame_seed, ref_seed=None):
    """One frame through the SD1.5 oracle; returns what the parity tests compare: the image, the TAESD-encoded latents,
    the final denoised latents."""
    from PIL import Image

    kw = dict(height=H, width=W, strength=0.6, steps=steps, seed=23, keep_trace=True)
    if ref_seed is not None:
        kw["ref_image"] = Image.fromarray(frame(H, W, seed=ref_seed), "RGB")
    else:
        kw.update(controlnet_scale=cn_scale, use_controlnet=cn)
    want = np.asarray(orc.infer(Image.fromarray(frame(H, W, seed=frame_seed), "RGB"), text[None].float(), **kw))
    return {"want": want, "init_latents": orc.trace["init_latents"][0].numpy(), "denoised": orc.trace["denoised"][-1][0].numpy()}


def sdxl_infer(orc, text, pooled, H, W, steps, frame_seed=2):
    from PIL import Image

    want = np.asarray(orc.infer(Image.fromarray(frame(H, W, seed=frame_seed), "RGB"), text[None].float(), height=H, width=W,
                                strength=0.6, steps=steps, seed=23, use_controlnet=False, keep_trace=True, pooled=pooled))
    return {"want": want, "denoised": orc.trace["denoised"][-1][0].numpy()}


SD15_JOBS = {  # name -> sd15_infer arguments after (orc, text)
    "sd15_config2": (512, 512, 4, True, 1.0, 31),
    "sd15_config5": (768, 768, 8, True, 2.0, 41),
    "sd15_ref512": (512, 512, 4, False, 1.0, 51, 52),
}
SDXL_JOBS = {"sdxl_1024": (1024, 1024, 4)}
JOBS = list(SD15_JOBS) + list(SDXL_JOBS)


def _build_and_run(name):
    """Child process: the same seeded weights as the test fixtures (synthesised on the GPU like theirs -- the per-tensor
    generators are device generators -- then moved to the host), the oracle, one call."""
    import torch

    from oracle.pipeline import OraclePipeline
    from videosd_amd import config as C
    from videosd_amd import weights as W

    cpu = lambda w: {k: v.cpu() for k, v in w.items()}  # noqa: E731
    wv = cpu(W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda"))
    if name in SD15_JOBS:
        wu = cpu(W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda"))
        wc = cpu(W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda"))
        text = (torch.randn(77, C.SD15_UNET.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5).half()
        torch.cuda.empty_cache()
        return sd15_infer(OraclePipeline(C.SD15_UNET, C.SD15_CONTROLNET, wu, wc, wv), text, *SD15_JOBS[name])
    cfg = C.SDXL_UNET
    wu = cpu(W.synthesize(W.unet_spec(cfg), "sdxl.", device="cuda"))
    g = torch.Generator().manual_seed(11)
    text = (torch.randn(77, cfg.cross_dim, generator=g) * 0.5).half()
    pooled = (torch.randn(cfg.add_pooled_dim, generator=g) * 0.5).half()
    torch.cuda.empty_cache()
    return sdxl_infer(OraclePipeline(cfg, None, wu, None, wv), text, pooled, *SDXL_JOBS[name])


# ---------------------------------------------------------------- the parent side
_children = {}  # name -> (Popen, path, t0)
_dir = None


def start_all(names=None, threads=None):
    """Start one child per job (idempotent).  Each gets a share of the host cores; the largest jobs first."""
    global _dir
    if os.environ.get("VSD_TEST_ORACLE_AHEAD", "1") == "0":
        return
    names = [n for n in (names or ["sdxl_1024", "sd15_config5", "sd15_config2", "sd15_ref512"]) if n not in _children]
    if not names:
        return
    if _dir is None:
        _dir = tempfile.mkdtemp(prefix="vsd_oracle_ahead_")
    cores = os.cpu_count() or 8
    per = threads or max(4, cores // max(1, len(names)))
    for n in names:
        env = dict(os.environ, OMP_NUM_THREADS=str(per), MKL_NUM_THREADS=str(per))
        path = os.path.join(_dir, n + ".npz")
        log = open(os.path.join(_dir, n + ".log"), "w")
        _children[n] = (subprocess.Popen([sys.executable, os.path.abspath(__file__), n, path], env=env, stdout=log, stderr=log,
                                         cwd=ROOT), path, time.time())


def result(name, timeout=1500.0):
    """The child's result (waits for it), or None when there is none (never started, failed, timed out): the caller then
    runs the oracle inline."""
    if name not in _children:
        return None
    proc, path, t0 = _children[name]
    try:
        proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        return None
    if proc.returncode != 0 or not os.path.exists(path):
        return None
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def stop_all():
    for proc, _, _ in _children.values():
        if proc.poll() is None:
            proc.kill()
    _children.clear()


if __name__ == "__main__":
    out = _build_and_run(sys.argv[1])
    np.savez(sys.argv[2], **out)
