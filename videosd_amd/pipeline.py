"""VideoSDPipeline — the reference's per-GPU worker class, re-hosted on the MI355X engine.

Mirrors /root/reference/diffusert/videopipeline.py:11-128: same constructor kwargs (`model`, `controlnet`
required, `device` optional, extras such as `gpus` / `compile` ignored), same `infer` signature and
defaults, same `load_model` / `compile_model` methods, so diffusert/server.py:104-117,317-321 can drive it
unchanged (`VideoSDPipeline.remote(**config)` and `await pipelines[gpu].infer.remote(img, **options)` are
provided by videosd_amd/dispatch.py without Ray).

What differs underneath: no diffusers / TensorRT / torch.compile; the frame goes through
videosd_amd.engine.Engine (libvsd.so HIP kernels replayed as one hipGraph).
"""
import os
import time
import zlib
from typing import Dict, List, Optional, Union

import numpy as np
import torch
from PIL import Image

from . import config as C
from . import checkpoints as CK
from . import weights as W


def center_crop_resize(img: Image.Image, width: int, height: int) -> Image.Image:
    """Center crop to the target aspect ratio (float box) and LANCZOS-resize: videopipeline.py:92-107."""
    if img.width / img.height > width / height:
        new_width = img.height * (width / height)
        box = ((img.width - new_width) / 2, 0, (img.width + new_width) / 2, img.height)
    else:
        new_height = img.width * (height / width)
        box = (0, (img.height - new_height) / 2, img.width, (img.height + new_height) / 2)
    return img.crop(box).resize((width, height), resample=Image.Resampling.LANCZOS)


def _weights_dir() -> Optional[str]:
    d = os.environ.get("VSD_WEIGHTS")
    return d if d and os.path.isdir(d) else None


def load_or_synthesize(spec, prefix: str, filename: str, device, snapshot_file: Optional[str] = None):
    """-> (weights, source).  In this order: the safetensors file of a `from_pretrained` directory (`snapshot_file`, found by
    weights.checkpoint_file), `$VSD_WEIGHTS/<filename>` (flat layout), seeded synthetic tensors (no checkpoint exists offline).
    A file that lacks a tensor of the architecture, or holds it in another shape, is refused with the tensor's name."""
    d = _weights_dir()
    path = snapshot_file or (os.path.join(d, filename) if d and os.path.exists(os.path.join(d, filename)) else None)
    if path is not None:
        w = CK.load_safetensors(path, device=device)
        CK.check_against_spec(w, spec, path if snapshot_file else filename)
        return w, path
    return W.synthesize(spec, prefix, device=device), "synthetic"


class VideoSDPipeline:
    """One instance = one GPU = one full weight replica (the reference's Ray actor with num_gpus=1)."""

    def __init__(self, *args, **kwargs):
        self.device = kwargs.get("device", 0)
        self._kwargs = dict(kwargs)
        self.honor_controlnet_flag = bool(kwargs.get("honor_controlnet_flag", False))  # extension, off by default
        # extension, off by default: `ref=True` runs the reference-only mode (lcm_reference_pipeline.py, dead code at the
        # reference's v2 where `ref` / `style_fidelity` are accepted and ignored, videopipeline.py:84-85)
        self.honor_ref_flag = bool(kwargs.get("honor_ref_flag", False))
        self._ref_img = None
        self._ref_epoch = 0
        # launch lanes this instance may keep in flight (`submit_batch(lane=...)`, the worker loop of dispatch.py): lane l runs on
        # launch stream l (ops.HipOps); with at most two lanes every lane also has a stream for its side branch
        self.max_lanes = max(1, int(kwargs.get("lanes", 2)))
        try:
            self.load_model(kwargs["model"], kwargs["controlnet"])
        except KeyError:
            print("Model name and controlnet model must be specified")  # videopipeline.py:24-26
            raise
        # Per-session state without stalls (server.py:90-93: options live on each VideoSDTrack; :132-137: every session's
        # frames go through the same actors).  Two LRU caches:
        #   prompts: prompt key -> PromptBlock (cross-attention K / V^T + absorbed weights of every layer, ~40 MB, built on the
        #            GPU in ~1 ms); an engine takes a cached prompt with one device-to-device copy on its own stream;
        #   plans:   (size, steps, timestep count, ControlNet, ref) -> {engines by (frames per launch, lane), constants}: each
        #            plan has its own schedule constants, so sessions with different sizes / steps alternate frame by frame
        #            with their graphs intact.  `max_plans` counts PROGRAMS.  Every engine owns its activation arena (1-6 GB
        #            at 512x512, one to five frames per launch): `memory_budget` (fraction of the device's memory, default
        #            0.6) bounds what the cache may hold -- least recently used idle programs go first, then the idle slots
        #            of the program being extended (`_trim_memory`); an engine that is needed again is prepared again.
        from collections import OrderedDict

        self._prompts = OrderedDict()
        self.max_prompts = int(kwargs.get("max_prompts", 8))
        self._plans = OrderedDict()
        self.max_plans = int(kwargs.get("max_plans", 3))
        self.memory_budget = float(kwargs.get("memory_budget", 0.6))
        self.evictions = 0
        self._outstanding = []  # engines with a submitted, not yet collected launch
        self._lanes_busy = []   # ... and the lane each of them runs on
        self._host_ms = {"crop_resize": [], "upload_enqueue": [], "wait_download": [], "to_pil": [], "gpu": [], "prepare": [], "update_options": [],
                         "prompt": []}

    # ------------------------------------------------------------------ model loading
    def load_model(self, model_name, controlnet_model="lllyasviel/control_v11p_sd15_canny"):
        """videopipeline.py:49-72.  Like the reference, `model_name` is accepted but the UNet is always the
        LCM-distilled SD1.5 UNet (the reference hard-codes SimianLuo/LCM_Dreamshaper_v7), the VAE is TAESD and
        the ControlNet is SD1.5-canny."""
        from .engine import Engine
        from .ops import HipOps  # raises without a ROCm GPU / libvsd.so: there is no CPU path

        ops = HipOps(int(self.device))
        dev = ops.device
        # per-shape kernel configurations found on an MI355X (shapes not in the table are timed at `prepare`)
        tuning = os.environ.get("VSD_TUNING") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "profiles", "tuning_mi355x.json")
        ops.load_tuning(tuning)
        # "auto": shapes missing from the table are timed on this GPU at `prepare`; "table": never -- missing shapes take the
        # deterministic heuristic (ops.choose_tile), so that every rank of a group builds the SAME kernels and a frame's bits
        # do not depend on the rank it lands on (dispatch.spawn_workers sets this for groups; `sync_tuning` spreads rank 0's
        # measured choices)
        self.tuning_mode = str(getattr(self, "_kwargs", {}).get("tuning_mode", "auto"))
        # Extension (BASELINE.json configs[3]; the reference only knows SD1.5): a model name containing "xl" selects the
        # SDXL-base UNet topology (LCM-SDXL weights as `unet_sdxl.safetensors`, TAESD-XL as `taesdxl.safetensors`),
        # without a ControlNet tower; prompts then need 2048-wide embeddings + a 1280-wide pooled vector.
        self.is_xl = "xl" in str(model_name).lower()
        self.unet_cfg = C.SDXL_UNET if self.is_xl else C.SD15_UNET
        # What `from_pretrained` reads (videopipeline.py:51-69), offline: `model` / `controlnet` given as DIRECTORIES in the
        # Hugging Face snapshot layout -- <model>/unet/diffusion_pytorch_model.safetensors, <model>/text_encoder/model.safetensors,
        # <model>/tokenizer/{vocab.json, merges.txt}; the ControlNet's and TAESD's diffusion_pytorch_model.safetensors at the top
        # level of theirs -- or hub ids with a snapshot in the local HF cache; fp32 or fp16 tensors, cast to fp16 like
        # torch_dtype=float16.  Then the flat $VSD_WEIGHTS/*.safetensors files, then seeded synthetic tensors.  Like the
        # reference, a hub id that is not SimianLuo/LCM_Dreamshaper_v7 still gets that UNet (it hard-codes the id, :57); the VAE is
        # `madebyollin/taesd` (:68) unless the extension kwarg `vae=` names another directory.
        kw = getattr(self, "_kwargs", {})
        mdir = CK.find_snapshot(model_name) or (None if self.is_xl else CK.find_snapshot("SimianLuo/LCM_Dreamshaper_v7"))
        vdir = CK.find_snapshot(kw.get("vae") or ("madebyollin/taesdxl" if self.is_xl else "madebyollin/taesd"))
        self.weight_sources = {}
        if self.is_xl:
            wu, self.weight_sources["unet"] = load_or_synthesize(W.unet_spec(C.SDXL_UNET), "sdxl.", "unet_sdxl.safetensors", dev,
                                                                 CK.checkpoint_file(mdir, "unet"))
            wv, self.weight_sources["vae"] = load_or_synthesize(W.taesd_spec(C.TAESD), "vae.", "taesdxl.safetensors", dev,
                                                                CK.checkpoint_file(vdir))
            self.model = Engine(ops, C.SDXL_UNET, None, C.TAESD, wu, None, wv)
            # the two text towers of an SDXL snapshot (<model>/text_encoder, <model>/text_encoder_2, tokenizer, tokenizer_2;
            # flat: $VSD_WEIGHTS/text_encoder.safetensors + text_encoder_2.safetensors, vocabulary files in $VSD_WEIGHTS[/tokenizer_2])
            self.text_encoder, self._xl_prompts = None, {}
            d = _weights_dir()

            def flat(name):
                return os.path.join(d, name) if d and os.path.exists(os.path.join(d, name)) else None

            def sub(name):
                return os.path.join(mdir, name) if mdir and os.path.isdir(os.path.join(mdir, name)) else None

            f1 = CK.checkpoint_file(mdir, "text_encoder", stems=("model",)) or flat("text_encoder.safetensors")
            f2 = CK.checkpoint_file(mdir, "text_encoder_2", stems=("model",)) or flat("text_encoder_2.safetensors")
            self.weight_sources["text_encoder"] = f1 or "stand-in embeddings seeded by the prompt text"
            self.weight_sources["text_encoder_2"] = f2 or "stand-in embeddings seeded by the prompt text"
            if f1 is not None and f2 is not None:
                from . import clip as K

                towers = []
                for f, cfg, tok in ((f1, K.SDXL_CLIP_L, sub("tokenizer") or d),
                                    (f2, K.SDXL_CLIP_G, sub("tokenizer_2") or (flat("tokenizer_2") or d))):
                    wt = CK.load_safetensors(f, device=dev)
                    CK.check_against_spec(wt, K.text_tower_spec(cfg), f)
                    towers.append(K.ClipTextEncoder(ops, cfg, wt, tokenizer_dir=tok))
                self.text_encoder = K.SdxlTextEncoders(*towers)
            return self.model
        cdir = CK.find_snapshot(controlnet_model)
        wu, self.weight_sources["unet"] = load_or_synthesize(W.unet_spec(C.SD15_UNET), "unet.", "unet.safetensors", dev,
                                                             CK.checkpoint_file(mdir, "unet"))
        wc, self.weight_sources["controlnet"] = load_or_synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", "controlnet.safetensors", dev,
                                                                   CK.checkpoint_file(cdir))
        wv, self.weight_sources["vae"] = load_or_synthesize(W.taesd_spec(C.TAESD), "vae.", "taesd.safetensors", dev,
                                                            CK.checkpoint_file(vdir))
        self.model = Engine(ops, C.SD15_UNET, C.SD15_CONTROLNET, C.TAESD, wu, wc, wv)
        self.text_encoder = None
        d = _weights_dir()
        clip_file = CK.checkpoint_file(mdir, "text_encoder", stems=("model",))
        tok_dir = os.path.join(mdir, "tokenizer") if mdir and os.path.isdir(os.path.join(mdir, "tokenizer")) else d
        if clip_file is None and d and os.path.exists(os.path.join(d, "text_encoder.safetensors")):
            clip_file = os.path.join(d, "text_encoder.safetensors")
        self.weight_sources["text_encoder"] = clip_file or "stand-in embeddings seeded by the prompt text"
        if clip_file is not None:
            from .clip import ClipTextEncoder

            wt = CK.load_safetensors(clip_file, device=dev)
            CK.check_against_spec(wt, W.clip_spec(C.CLIP_L), clip_file)
            self.text_encoder = ClipTextEncoder(ops, C.CLIP_L, wt, tokenizer_dir=tok_dir)
        return self.model

    def compile_model(self):
        """videopipeline.py:35-47: build the replayable graph and run the 768x768 warm-up frame."""
        return self.infer(Image.new("RGB", (768, 768)), prompt="warmup", height=768, width=768, strength=0.8, steps=4)

    # ------------------------------------------------------------------ prompt
    def encode_prompt(self, prompt: Union[str, List[str]]) -> torch.Tensor:
        """[77, 768] fp16 embeddings.  With CLIP weights + tokenizer files under $VSD_WEIGHTS the HIP CLIP encoder
        runs; otherwise (no vocabulary offline) a deterministic stand-in seeded by the prompt text is used."""
        text = self._one_prompt(prompt)
        if self.text_encoder is not None and self.text_encoder.has_tokenizer:
            return self._xl_encode(text)[0] if self.is_xl else self.text_encoder.encode(text)
        g = torch.Generator().manual_seed(zlib.crc32(text.encode()))
        return (torch.randn(77, self.unet_cfg.cross_dim, generator=g) * 0.5).half()

    @staticmethod
    def _one_prompt(prompt: Union[str, List[str]]) -> str:
        """The ONE prompt of a frame.  The reference hands a list to the tokenizer as a BATCH: `batch_size = len(prompt)`
        (lcm_controlnet.py:433-438), i.e. a two-element list asks for two images from one input frame -- which its own
        `infer` cannot return (videopipeline.py:126-128 takes `.images[0]`) and `prepare_latents` cannot build from a single
        encoded frame without duplicating it.  Every call site of the reference passes a string or the one-element default
        `["pixar, cg"]` (videopipeline.py:77; server.py:166-171 sets a string): that is what is supported here, and anything
        else is refused instead of silently meaning something different (rounds 1-4 joined the elements with spaces)."""
        if isinstance(prompt, str):
            return prompt
        items = list(prompt)
        if len(items) != 1 or not isinstance(items[0], str):
            raise ValueError(f"prompt must be a string or a one-element list of strings, got {len(items)} element(s): the reference treats "
                             "a list as a batch of prompts (batch_size = len(prompt), lcm_controlnet.py:433-438), which this per-frame "
                             "path (one image in, one image out, videopipeline.py:126-128) does not have")
        return items[0]

    def _xl_encode(self, text: str):
        """SDXL: (prompt_embeds [77, 2048], pooled_prompt_embeds [1280]) of both text towers, once per prompt text (the pooled
        vector is asked for separately, per engine: keep the last few pairs)."""
        hit = self._xl_prompts.get(text)
        if hit is None:
            hit = self._xl_prompts[text] = self.text_encoder.encode(text)
            while len(self._xl_prompts) > 8:
                self._xl_prompts.pop(next(iter(self._xl_prompts)))
        return hit

    def encode_pooled(self, prompt: Union[str, List[str]]) -> torch.Tensor:
        """SDXL only: the pooled text embedding [1280] = the second tower's `text_embeds` (with both towers' checkpoints and
        tokenizer files present; otherwise a stand-in seeded by the prompt text, like `encode_prompt`)."""
        text = self._one_prompt(prompt)
        if self.text_encoder is not None and self.text_encoder.has_tokenizer:
            return self._xl_encode(text)[1]
        g = torch.Generator().manual_seed(zlib.crc32(("pooled:" + text).encode()))
        return (torch.randn(self.unet_cfg.add_pooled_dim, generator=g) * 0.5).half()

    @property
    def prompt_shape(self):
        """Shape of the embeddings `set_prompt_embeds` takes (what the RCCL broadcast of dispatch.py carries)."""
        return (77, self.unet_cfg.cross_dim)

    def set_prompt_embeds(self, embeds: torch.Tensor, key=None):
        """Install embeddings produced elsewhere (rank 0 broadcasts them over RCCL, dispatch.py) as prompt `key`: they join
        the prompt cache; nothing in flight is disturbed."""
        return self._cache_prompt(key, embeds)

    def _cache_prompt(self, key, embeds=None, prompt=None):
        blk = self._prompts.get(key)
        if blk is not None and embeds is not None and blk.text is not None:
            # the same embeddings for a key this worker already holds (a repeated sync, a dispatcher that lost track of what the
            # workers know): keep the block -- rebuilding it is 23 K / V GEMMs + 16 folds + a synchronise, and every engine
            # would re-copy its 40 MB at its next launch
            e = embeds.reshape(-1, embeds.shape[-1]).to(device=blk.text.device, dtype=torch.float16)
            if e.shape == blk.text.shape and bool(torch.equal(e, blk.text)):
                embeds = None
        if blk is None or embeds is not None:
            t0 = time.perf_counter()
            blk = self.model.build_prompt(embeds if embeds is not None else self.encode_prompt(prompt))
            self._prompts[key] = blk
            self._note("prompt", t0)
            while len(self._prompts) > max(1, self.max_prompts):  # least recently used first; an engine that still holds an
                self._prompts.popitem(last=False)                 # evicted block keeps it alive until it switches
        self._prompts.move_to_end(key)
        return blk

    def set_reference(self, img):
        """The reference image of the reference-only mode (`honor_ref_flag=True`, `infer(..., ref=True)`).  Without one the
        first frame seen with `ref=True` becomes the reference."""
        self._require_idle("change the reference image")
        self._ref_img = img
        self._ref_epoch += 1

    def can_batch(self, **options) -> bool:
        """May the worker coalesce queued frames with these options into one launch?  (reference-only frames go alone)"""
        return not (self.honor_ref_flag and options.get("ref", False))

    def _require_idle(self, what: str):
        """Prompt constants (cross-attention K / V^T) and plans are shared by every lane: rewriting them under a launch
        that is still running would hand that launch's caller another plan's frame."""
        if self._outstanding:
            raise RuntimeError(f"cannot {what} while {len(self._outstanding)} launch(es) are in flight: collect them first")

    def _note(self, what: str, t0: float):
        v = self._host_ms[what]
        v.append((time.perf_counter() - t0) * 1e3)
        if len(v) > 256:
            del v[:128]

    def metrics(self):
        """Per-stage host / device milliseconds of this worker (medians over the last frames): where a frame's time goes
        between `infer` entry and the PIL image (per-stage ms per GPU)."""
        import statistics

        out = {k: (round(statistics.median(v), 3) if v else None) for k, v in self._host_ms.items()}
        out["plans_cached"] = len(self._engines)
        out["programs_cached"] = len(self._plans)
        out["prompts_cached"] = len(self._prompts)
        out["engines_evicted_for_memory"] = self.evictions
        free, total = torch.cuda.mem_get_info(int(self.device))  # what a leak across plan / prompt changes would show in
        return {"stage_ms_p50": out, "device_free_mb": free >> 20, "device_total_mb": total >> 20,
                "allocated_mb": torch.cuda.memory_allocated(int(self.device)) >> 20}

    def stage_profile(self, batch: int = 1):
        """Per-kernel-family device ms of ONE eager pass of the current plan (vsd_stage_times; the captured graph cannot be
        bracketed per launch).  On demand only: it re-runs the program un-captured."""
        self._require_idle("profile")
        if not self._plans:
            raise RuntimeError("no plan prepared yet: call infer first")
        eng = next(reversed(self._plans.values()))["root"]  # the most recently used program
        ops = eng.ops
        eng.program.run()
        ops.synchronize()
        ops.profile_begin()
        eng.program.run()
        ops.synchronize()
        return {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in ops.profile_end().items()}

    # ------------------------------------------------------------------ per frame
    def infer(
        self,
        img,
        prompt=["pixar, cg"],
        height=360,
        width=640,
        strength=0.4,
        steps=20,
        guidance_scale=7.5,
        ref=False,
        style_fidelity=0.0,
        controlnet=False,
        seed=42,
        controlnet_scale=1,
    ):
        """Same contract as videopipeline.py:75-128.  `guidance_scale`, `ref`, `style_fidelity` and
        `controlnet` are accepted and ignored exactly like the reference (ControlNet always runs; 7.5 is baked
        in); `seed` does not change the result because the reference resets the CPU generator state per frame."""
        return self.infer_batch([img], prompt=prompt, height=height, width=width, strength=strength, steps=steps,
                                guidance_scale=guidance_scale, ref=ref, style_fidelity=style_fidelity,
                                controlnet=controlnet, seed=seed, controlnet_scale=controlnet_scale)[0]

    def infer_batch(self, imgs, prompt=["pixar, cg"], height=360, width=640, strength=0.4, steps=20, guidance_scale=7.5,
                    ref=False, style_fidelity=0.0, controlnet=False, seed=42, controlnet_scale=1):
        """Several frames (of different sessions, or consecutive frames of one stream) with the SAME options through
        one batched launch: the result `infer` gives each frame, up to kernel rounding (frames are denoised independently), one pass over the
        weights for all of them.  Extension of the reference surface; `RemotePipeline(batch=B)` coalesces queued
        `infer` calls into this."""
        return self.collect_batch(self.submit_batch(imgs, prompt=prompt, height=height, width=width, strength=strength,
                                                    steps=steps, guidance_scale=guidance_scale, ref=ref,
                                                    style_fidelity=style_fidelity, controlnet=controlnet, seed=seed,
                                                    controlnet_scale=controlnet_scale))

    def warm_up(self, batches=(1,), lanes: int = 1, **options):
        """Prepare every (batch size, lane) engine a stream with these `infer` options will use -- plan, captured graph
        and one replay on black frames -- so that no frame of the stream pays for a `prepare` (tens of ms each, a
        visible stall in a 30 ms/frame stream).  A server calls it once per worker at start-up or after a respawn
        (`FrameDispatcher(warm_options=...)`); returns how many engines are ready."""
        w, h = options.get("width", 640), options.get("height", 360)
        black = Image.new("RGB", (w, h))
        ready = 0
        for b in batches:
            handles = [self.submit_batch([black] * int(b), lane=l, **options) for l in range(max(1, int(lanes)))]
            for hd in handles:
                self.collect_batch(hd)
            ready += len(handles)
        return ready

    def submit_batch(self, imgs, lane: int = 0, prompt=["pixar, cg"], height=360, width=640, strength=0.4, steps=20,
                     guidance_scale=7.5, ref=False, style_fidelity=0.0, controlnet=False, seed=42, controlnet_scale=1):
        """First half of `infer_batch`: crop / resize, upload, enqueue -- returns a handle for `collect_batch` without
        waiting for the GPU.  `lane` picks one of the prepared engines of that (options, batch size): two lanes keep two
        launches in flight while the host works on the frames around them (the worker loop of dispatch.py does that)."""
        if not 0 <= int(lane) < self.max_lanes:
            raise ValueError(f"lane {lane}: this pipeline was built for {self.max_lanes} launch lane(s) (kwarg `lanes`)")
        t0 = time.perf_counter()
        imgs = [center_crop_resize(im, width, height) for im in imgs]
        # A size that is not a multiple of the VAE stride: `VaeImageProcessor.preprocess` (lcm_controlnet.py:457, 230) rounds it
        # DOWN to a multiple of 8 with a Lanczos resize and the pipeline returns that size.  (There the control image is the
        # Sobel map of the unrounded frame, resized; here it is the Sobel map of the rounded frame -- sizes the client offers
        # are multiples of 8.)
        if height % 8 or width % 8:
            height, width = height - height % 8, width - width % 8
            if height <= 0 or width <= 0:
                raise ValueError("height and width must be at least 8")
            imgs = [im.resize((width, height), resample=Image.Resampling.LANCZOS) for im in imgs]
        self._note("crop_resize", t0)
        pkey = prompt if isinstance(prompt, str) else tuple(prompt)
        pblock = self._cache_prompt(pkey, prompt=prompt)  # cached: nothing to do; new: ~1 ms on the GPU, nobody waits
        use_cn = False if self.is_xl else (bool(controlnet) if self.honor_controlnet_flag else True)
        use_ref = bool(ref) and self.honor_ref_flag and not self.is_xl
        if use_ref:
            if len(imgs) != 1:
                raise ValueError("ref=True: one frame per call (the reference-only program runs one frame per launch)")
            use_cn = False  # the reference-only pipeline has no ControlNet (lcm_reference_pipeline.py:855-890)
            if self._ref_img is None:
                self._ref_img = imgs[0]
                self._ref_epoch += 1
        # The captured program depends on the frame size, the NUMBER of timesteps and the ControlNet switch; `strength`
        # and `controlnet_scale` only change constants the graph reads from device memory (Engine.update_options): a
        # slider drag in the client (server.py:163-197) does not rebuild or re-capture anything.
        from .lcm import lcm_timesteps

        n_eff = len(lcm_timesteps(float(strength), int(steps)))  # ValueError for an empty schedule: the caller's problem
        # (SDXL: the pooled text embedding is baked into the time embeddings at `prepare`, so the prompt is part of the program)
        plan_key = (height, width, int(steps), n_eff, use_cn, use_ref) + ((pkey,) if self.is_xl else ())
        opts = (float(strength), float(controlnet_scale))
        eng = self._engine_for(plan_key, opts, len(imgs), lane, prompt=pblock, prompt_text=prompt)
        eng.use_prompt(pblock)  # this lane's launch reads ITS copy of the constants: the other lanes may run other prompts
        if use_ref and getattr(eng, "_ref_epoch", None) != self._ref_epoch:
            rf = np.asarray(center_crop_resize(self._ref_img.convert("RGB"), width, height), dtype=np.uint8)
            eng.ops.upload(eng.ref_u8, torch.from_numpy(np.array(rf, copy=True)))  # (PIL's buffer is read-only)
            eng._ref_epoch = self._ref_epoch
        np.random.seed(seed)  # kept for parity with videopipeline.py:112 (nothing downstream consumes it)
        t0 = time.perf_counter()
        frames = np.stack([np.asarray(im if im.mode == "RGB" else im.convert("RGB"), dtype=np.uint8) for im in imgs])
        eng.submit_u8(frames[0] if len(imgs) == 1 else frames, overlap=self._overlap_now(lane))
        self._outstanding.append(eng)
        self._lanes_busy.append(int(lane))
        self._note("upload_enqueue", t0)
        return (eng, len(imgs))

    def _overlap_now(self, lane: int) -> bool:
        """ONE policy for every caller (bench.py's engine legs follow the same rule): a launch runs its ControlNet encoder on
        the lane's side stream when it will have a command-processor pipe to itself -- at most two launches in flight, all on
        lanes 0 / 1 (lane l's side stream is lane l + 2's own stream).  A lone frame gains ~4 ms from it; with three or four
        lanes busy the side streams ARE the other lanes' streams and the launch stays on its own."""
        if os.environ.get("VSD_OVERLAP_CN") is not None:
            return os.environ.get("VSD_OVERLAP_CN") == "1"
        return int(lane) < 2 and all(l < 2 for l in self._lanes_busy) and len(self._lanes_busy) < 2

    def collect_batch(self, handle):
        eng, n = handle
        t0 = time.perf_counter()
        try:
            out = eng.collect_u8()
        finally:
            if eng in self._outstanding:
                i = self._outstanding.index(eng)
                self._outstanding.pop(i)
                self._lanes_busy.pop(i)
        self._note("wait_download", t0)
        if getattr(eng, "last_gpu_ms", None) is not None:
            self._host_ms["gpu"].append(eng.last_gpu_ms)
        out = out[None] if n == 1 else out
        t0 = time.perf_counter()
        res = [Image.fromarray(o, mode="RGB") for o in out]
        self._note("to_pil", t0)
        return res

    @property
    def _engines(self):
        """(plan_key, frames per launch, lane) -> prepared engine, over all cached plans"""
        return {(pk, b, l): e for pk, pl in self._plans.items() for (b, l), e in pl["engines"].items()}

    def _plan_busy(self, plan) -> bool:
        return any(e in self._outstanding for e in plan["engines"].values())

    def needs_idle(self, **options) -> bool:
        """Would a frame with these `infer` options have to wait for launches in flight?  Only a change of `strength` /
        `controlnet_scale` of a plan that is running does (its graphs read those constants); other prompts, sizes, step
        counts and batch sizes go beside what is running.  (dispatch.py's worker drains before such a frame.)"""
        try:
            from .lcm import lcm_timesteps

            o = dict(height=360, width=640, strength=0.4, steps=20, controlnet=False, ref=False, controlnet_scale=1, prompt=["pixar, cg"])
            o.update(options)
            n_eff = len(lcm_timesteps(float(o["strength"]), int(o["steps"])))
            use_cn = False if self.is_xl else (bool(o["controlnet"]) if self.honor_controlnet_flag else True)
            use_ref = bool(o["ref"]) and self.honor_ref_flag and not self.is_xl
            if use_ref:
                return bool(self._outstanding)  # (reference image upload + one frame per launch: keep it simple)
            pk = (o["height"] - o["height"] % 8, o["width"] - o["width"] % 8, int(o["steps"]), n_eff, use_cn and not use_ref, use_ref)
            if self.is_xl:
                pk += (o["prompt"] if isinstance(o["prompt"], str) else tuple(o["prompt"]),)
            plan = self._plans.get(pk)
            if plan is None:
                return False
            return plan["opts"] != (float(o["strength"]), float(o["controlnet_scale"])) and self._plan_busy(plan)
        except Exception:
            return bool(self._outstanding)

    def _engine_for(self, plan_key, opts, batch: int, lane: int = 0, prompt=None, prompt_text=None):
        """A prepared engine per (program, frames per launch, lane).  A program's first engine owns its schedule constants;
        the others are slots of it (shared weights and constants, own arena / prompt constants / graph), so that switching
        between batch sizes, lanes, prompts and PROGRAMS costs nothing per frame.  `opts` = (strength, controlnet_scale): a
        change rewrites that program's device constants, nothing else."""
        height, width, steps, _n, use_cn, use_ref = plan_key[:6]
        strength, cn_scale = opts
        plan = self._plans.get(plan_key)
        if plan is not None:
            self._plans.move_to_end(plan_key)
            if opts != plan["opts"]:
                if self._plan_busy(plan):
                    raise RuntimeError("cannot change strength / controlnet_scale of a plan with launches in flight: collect them first")
                t0 = time.perf_counter()
                if not plan["root"].update_options(strength, cn_scale):  # (cannot happen: the timestep count is in the key)
                    raise RuntimeError("update_options refused a schedule with the same number of timesteps")
                plan["opts"] = opts
                self._note("update_options", t0)
            eng = plan["engines"].get((batch, lane))
            if eng is not None:
                if eng in self._outstanding:
                    raise RuntimeError("this lane's previous launch has not been collected")
                return eng
        t0 = time.perf_counter()
        self._trim_memory(keep=plan_key)
        if plan is None:
            # least recently used PROGRAMS go first, but never one with a launch still running (the cache then grows)
            for pk in list(self._plans):
                if len(self._plans) < max(1, self.max_plans):
                    break
                if not self._plan_busy(self._plans[pk]):
                    for e in self._plans.pop(pk)["engines"].values():
                        e._destroy_graphs()
            eng = self.model.make_slot(share_plan=False, lane=lane)  # its own schedule constants: the other programs keep running
            plan = self._plans[plan_key] = {"root": eng, "opts": opts, "engines": {}}
        else:
            eng = plan["root"].make_slot(lane=lane)
        if prompt is not None:
            eng.use_prompt(prompt)
        if self.is_xl:  # micro-conditioning: original size = target size = the frame size, no crop
            eng.set_added_cond(self.encode_pooled(prompt_text if prompt_text is not None else ""), (height, width, 0, 0, height, width))
        # (every engine is captured both ways -- ControlNet encoder on the lane's side stream / everything on the lane's own
        #  stream; `submit_batch` picks per launch, see `_overlap_now`)
        # Kernel choices: coalesced launches of a worker with three or four lanes are the loaded case -- the forms that cost least
        # with four lanes busy; one-frame launches and workers with one or two lanes are the latency case -- the forms that are
        # fastest alone.  Fixed per plan, the same on every lane (same bits whichever lane a frame lands on).
        eng.tune_for_lanes = self.max_lanes >= 3 and batch > 1
        eng.prepare(height, width, steps, strength, controlnet_scale=cn_scale, use_controlnet=use_cn, batch=batch, ref_mode=use_ref,
                    autotune=self.tuning_mode != "table")
        eng._ref_epoch = None
        plan["engines"][(batch, lane)] = eng
        self._note("prepare", t0)
        return eng

    def _memory_used_and_limit(self, before=None):
        dev = int(self.device)
        used = torch.cuda.memory_allocated(dev)
        if before is not None and used >= before:  # an evicted engine still referenced from a cycle: collect it now
            import gc

            gc.collect()
            used = torch.cuda.memory_allocated(dev)
        return used, self.memory_budget * torch.cuda.mem_get_info(dev)[1]

    def _trim_memory(self, keep=None):
        """Before another engine (= another arena) is allocated: while the allocator holds more than the budget, drop idle
        cached state, cheapest to rebuild last -- other programs (least recently used first), then the idle slots of the
        program `keep` itself (never its root: it owns the schedule constants the slots share).  Engines with a launch in
        flight are never touched; if nothing idle is left the allocation goes ahead (and may fail loudly in the allocator)."""
        used, limit = self._memory_used_and_limit()
        if used <= limit:
            return 0
        dropped = 0
        for pk in list(self._plans):
            if pk == keep or self._plan_busy(self._plans[pk]):
                continue
            for e in self._plans.pop(pk)["engines"].values():
                e._destroy_graphs()
                dropped += 1
            e = None
            used = self._memory_used_and_limit(before=used)[0]
            if used <= limit:
                break
        plan = self._plans.get(keep)
        if plan is not None and used > limit:
            for k, e in list(plan["engines"].items()):
                if e is plan["root"] or e in self._outstanding:
                    continue
                del plan["engines"][k]
                e._destroy_graphs()
                dropped += 1
                e = None
                used = self._memory_used_and_limit(before=used)[0]
                if used <= limit:
                    break
        self.evictions += dropped
        return dropped

    def export_plan(self, path: str, frames_per_launch: int = 1, **options) -> dict:
        """The program `infer(img, **options)` runs on this pipeline -- model, prompt, frame size, steps, strength, ControlNet scale --
        as a FILE for hosts without Python: include/vsd.h vsd_plan_load / vsd_plan_infer (examples/plan_host.c) replay it with the
        same kernels and arguments, bit for bit this class's frames.  The host feeds frames already cropped / resized to
        (height, width) (videopipeline.py:92-107 does that with PIL); another size, prompt or step count is another plan.
        `frames_per_launch` > 1: the coalesced program of `infer_batch`.  Returns the exporter's summary (videosd_amd/plan.py)."""
        from .plan import export_plan

        self._require_idle("export a plan")
        w, h = int(options.get("width", 640)), int(options.get("height", 360))
        imgs = [Image.new("RGB", (w, h), (127, 127, 127)) for _ in range(int(frames_per_launch))]
        handle = self.submit_batch(imgs, **options)
        self.collect_batch(handle)
        return export_plan(handle[0], path)

    def export_prompt(self, path: str, prompt) -> int:
        """Another prompt's constants as a file for vsd_plan_load_prompt (a loaded plan takes it without a new plan)."""
        from .plan import export_prompt

        pkey = prompt if isinstance(prompt, str) else tuple(prompt)
        return export_prompt(self._cache_prompt(pkey, prompt=prompt), path)

    def set_tuning_mode(self, mode: str):
        """ "auto" (time the candidates of shapes the table lacks at `prepare`) or "table" (never: deterministic heuristic)."""
        if mode not in ("auto", "table"):
            raise ValueError("tuning_mode must be 'auto' or 'table'")
        self.tuning_mode = mode
        return mode

    def export_tuning(self):
        """The per-shape kernel choices of this process (table + what `prepare` measured), for `import_tuning` elsewhere."""
        return {k: tuple(v) for k, v in self.model.ops.tile_override.items()}

    def import_tuning(self, table):
        """Take another process's choices for shapes this one has none for yet (rank 0's after its warm-up: every rank then
        builds the same kernels for the same shapes -- same bits whichever rank a frame lands on)."""
        n = 0
        for k, v in table.items():
            if k not in self.model.ops.tile_override:
                self.model.ops.tile_override[k] = tuple(v)
                n += 1
        return n

    # `VideoSDPipeline.remote(**config)` -> awaitable handle (replaces the Ray actor API, server.py:320-321)
    @classmethod
    def remote(cls, **config):
        from .dispatch import RemotePipeline

        return RemotePipeline(**config)

    @classmethod
    def remote_group(cls, gpus: int, **config):
        """`gpus` workers, one per GPU, in one RCCL process group (config.yaml `gpus: N`, server.py:273-274, 317-321)."""
        from .dispatch import spawn_workers

        return spawn_workers(int(gpus), **config)


VideoPipeline = VideoSDPipeline  # BASELINE.json's north_star calls the class by this name
