"""TEST-ONLY pipeline stand-ins used to exercise videosd_amd.dispatch without a GPU."""
import time

import numpy as np
from PIL import Image


class FakePipeline:
    """Same `infer` surface as VideoSDPipeline; 'diffusion' = invert the image after an optional delay."""

    def __init__(self, **config):
        if "model" not in config or "controlnet" not in config:
            raise KeyError("model")
        self.device = config.get("device", 0)
        self.delay = float(config.get("delay", 0.0))
        self.prompt = None

    def infer(self, img, prompt=["pixar, cg"], height=360, width=640, strength=0.4, steps=20, guidance_scale=7.5, ref=False,
              style_fidelity=0.0, controlnet=False, seed=42, controlnet_scale=1):
        if strength < 0:
            raise ValueError("negative strength")
        time.sleep(self.delay)
        a = 255 - np.asarray(img.convert("RGB").resize((width, height)))
        a[0, 0, 0] = self.device  # tag the worker that produced the frame
        return Image.fromarray(a.astype(np.uint8), "RGB")

    def infer_batch(self, imgs, **opts):
        """One 'launch' for all frames: one delay, every frame tagged with the size of the batch it rode in."""
        if opts.get("strength", 0.4) < 0:
            raise ValueError("negative strength")
        time.sleep(self.delay)
        outs = []
        for img in imgs:
            a = 255 - np.asarray(img.convert("RGB").resize((opts.get("width", 640), opts.get("height", 360))))
            a[0, 0, 0] = self.device
            a[0, 0, 1] = len(imgs)
            outs.append(Image.fromarray(a.astype(np.uint8), "RGB"))
        return outs

    # two-phase form (what the pipelined worker loop uses): the 'GPU work' is a timer started at submit
    def submit_batch(self, imgs, lane=0, **opts):
        if opts.get("strength", 0.4) < 0:
            raise ValueError("negative strength")
        self.lanes_used = getattr(self, "lanes_used", set()) | {lane}
        return (time.time() + self.delay, imgs, opts, lane)

    def collect_batch(self, handle):
        ready, imgs, opts, lane = handle
        time.sleep(max(0.0, ready - time.time()))
        outs = []
        for img in imgs:
            a = 255 - np.asarray(img.convert("RGB").resize((opts.get("width", 640), opts.get("height", 360))))
            a[0, 0, 0] = self.device
            a[0, 0, 1] = len(imgs)
            a[0, 0, 2] = lane
            outs.append(Image.fromarray(a.astype(np.uint8), "RGB"))
        return outs

    def compile_model(self):
        return Image.new("RGB", (8, 8))

    def set_prompt_embeds(self, embeds, key=None):
        self.prompt = float(embeds.float().sum())
        return self.prompt
