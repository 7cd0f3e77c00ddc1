"""TEST-ONLY pipeline stand-ins used to exercise videosd_amd.dispatch without a GPU."""
import os
import time
import zlib

import numpy as np
import torch
from PIL import Image


class FakePipeline:
    """Same `infer` surface as VideoSDPipeline; 'diffusion' = invert the image after an optional delay."""

    def __init__(self, **config):
        if "model" not in config or "controlnet" not in config:
            raise KeyError("model")
        self.device = config.get("device", 0)
        self.delay = float(config.get("delay", 0.0))
        self.prompt = None
        self.prompt_key = None
        self.prompt_shape = (77, 32)
        self.encodes = 0
        # fault injection: a frame whose first pixel's red value equals `crash_on` kills the process mid-frame
        # (a HIP fault / OOM kill looks like this from outside), `hang_on` makes it never answer (a hung GPU)
        self.crash_on = config.get("crash_on")
        self.hang_on = config.get("hang_on")

    def _faults(self, img):
        v = int(np.asarray(img.convert("RGB"))[0, 0, 0])
        if self.crash_on is not None and v == self.crash_on:
            os._exit(17)
        if self.hang_on is not None and v == self.hang_on:
            time.sleep(3600)

    def encode_prompt(self, prompt):
        """Deterministic stand-in embeddings; counts calls so tests can see WHICH rank encoded."""
        self.encodes += 1
        text = prompt if isinstance(prompt, str) else " ".join(prompt)
        g = torch.Generator().manual_seed(zlib.crc32(text.encode()))
        return torch.randn(*self.prompt_shape, generator=g).half()

    def prompt_state(self):
        return {"checksum": self.prompt, "key": self.prompt_key, "encodes": self.encodes}

    def infer(self, img, prompt=["pixar, cg"], height=360, width=640, strength=0.4, steps=20, guidance_scale=7.5, ref=False,
              style_fidelity=0.0, controlnet=False, seed=42, controlnet_scale=1):
        if strength < 0:
            raise ValueError("negative strength")
        if steps == 99:
            raise RuntimeError("HIP error: simulated device fault")
        self._faults(img)
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
        for im in imgs:
            self._faults(im)
        self.lanes_used = getattr(self, "lanes_used", set()) | {lane}
        imgs = [im.copy() for im in imgs]  # (the real pipeline has uploaded the pixels when submit returns)
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
        self.prompt_key = key
        return self.prompt


class OrderCheckingPipeline(FakePipeline):
    """Raises if a launch with new options is submitted while a launch with other options is still uncollected --
    what VideoSDPipeline's `_require_idle` enforces on the real engine."""

    def __init__(self, **config):
        super().__init__(**config)
        self.live = []

    def submit_batch(self, imgs, lane=0, **opts):
        key = (opts.get("strength"), opts.get("prompt"))
        if any(k != key for k in self.live):
            raise RuntimeError("options changed under a launch in flight")
        self.live.append(key)
        return super().submit_batch(imgs, lane=lane, **opts)

    def collect_batch(self, handle):
        self.live.pop(0)
        return super().collect_batch(handle)


class WarmFakePipeline(FakePipeline):
    """A stand-in that has `warm_up`, like VideoSDPipeline: records what it was warmed with."""

    def warm_up(self, batches=(1,), lanes=1, **options):
        self.warmed = {"batches": tuple(batches), "lanes": lanes, "options": sorted(options)}
        return len(tuple(batches)) * lanes

    def warm_state(self):
        return getattr(self, "warmed", None)


class SessionFakePipeline(FakePipeline):
    """Like VideoSDPipeline since round 3: plans and prompts are cached per key, launches of different sessions may be in
    flight together (`needs_idle` only asks for a drain when strength changes under a running plan).  Records how many
    launches of DIFFERENT sessions were live at once, and the tuning tables it exported / imported."""

    def __init__(self, **config):
        super().__init__(**config)
        self.live = []
        self.max_mixed = 0
        self.prompts = {}
        self.tuning = {("shape", self.device): ("tile", self.device)}

    def needs_idle(self, **opts):
        key = (opts.get("prompt"), opts.get("width"), opts.get("height"))
        return any(k == key and s != opts.get("strength") for k, s in self.live)

    def submit_batch(self, imgs, lane=0, **opts):
        key = (opts.get("prompt"), opts.get("width"), opts.get("height"))
        if any(k == key and s != opts.get("strength") for k, s in self.live):
            raise RuntimeError("strength changed under a running plan")
        self.live.append((key, opts.get("strength")))
        self.max_mixed = max(self.max_mixed, len({k for k, _ in self.live}))
        return super().submit_batch(imgs, lane=lane, **opts)

    def collect_batch(self, handle):
        self.live.pop(0)
        return super().collect_batch(handle)

    def set_prompt_embeds(self, embeds, key=None):
        self.prompts[key] = float(embeds.float().sum())
        return super().set_prompt_embeds(embeds, key=key)

    def session_state(self):
        return {"max_mixed": self.max_mixed, "prompts": sorted(self.prompts), "encodes": self.encodes, "tuning": dict(self.tuning)}

    def export_tuning(self):
        return dict(self.tuning)

    def import_tuning(self, table):
        n = 0
        for k, v in table.items():
            if k not in self.tuning:
                self.tuning[k] = v
                n += 1
        return n

    def warm_up(self, batches=(1,), lanes=1, **options):
        self.tuning.setdefault(("warmed", tuple(batches), lanes), ("by", self.device))  # a shape this rank "measures" if it has no entry yet
        return len(tuple(batches)) * lanes


class NullPipeline(FakePipeline):
    """A zero-cost 'GPU': every frame's result is one cached image of the asked size (no per-frame pixel work in the worker), so
    that what remains is the transport -- shared-memory slots, the request / reply pipes, the parent's threads and event loop
    (scripts/dispatch_ceiling.py: the host-side ceiling of the N-worker product path)."""

    def __init__(self, **config):
        super().__init__(**config)
        self._out = {}

    def _image(self, opts):
        key = (opts.get("width", 640), opts.get("height", 360))
        img = self._out.get(key)
        if img is None:
            a = np.zeros((key[1], key[0], 3), dtype=np.uint8)
            a[0, 0, 0] = self.device if isinstance(self.device, int) else 0
            img = self._out[key] = Image.fromarray(a, "RGB")
        return img

    def infer(self, img, **opts):
        return self._image(opts)

    def infer_batch(self, imgs, **opts):
        return [self._image(opts)] * len(imgs)

    def submit_batch(self, imgs, lane=0, **opts):
        return (len(imgs), opts)

    def collect_batch(self, handle):
        n, opts = handle
        return [self._image(opts)] * n
