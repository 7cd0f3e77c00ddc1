"""Per-frame algorithm restated on CPU fp32 (TEST INFRASTRUCTURE, see oracle/__init__).

Follows /root/reference/diffusert/videopipeline.py:75-128 (`VideoSDPipeline.infer`),
diffusert/lcm/canny_gpu.py:27-44 (`SobelOperator.forward`) and
diffusert/lcm/lcm_controlnet.py:379-618 (`LatentConsistencyModelPipeline_controlnet.__call__`).
The tokenizer vocabulary is not available offline, so the prompt enters as token ids or as
ready-made embeddings `[1,77,cross_dim]`.
"""
import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from . import nets
from .scheduler import LCMSchedulerOracle, w_embedding


def center_crop_resize(img: Image.Image, width: int, height: int) -> Image.Image:
    """videopipeline.py:92-107: float crop box to the target aspect, then LANCZOS resize."""
    if img.width / img.height > width / height:
        new_w = img.height * (width / height)
        box = ((img.width - new_w) / 2, 0, (img.width + new_w) / 2, img.height)
    else:
        new_h = img.width * (height / width)
        box = (0, (img.height - new_h) / 2, img.width, (img.height + new_h) / 2)
    return img.crop(box).resize((width, height), resample=Image.Resampling.LANCZOS)


def sobel_edges(img: Image.Image, low: float = 0.11, high: float = 0.8) -> Image.Image:
    """canny_gpu.py:27-44.  Returns a PIL 'L' image (ToPILImage: mul(255).byte() truncation)."""
    gray = np.asarray(img.convert("L"), dtype=np.float32) / 255.0
    x = torch.from_numpy(gray)[None, None]
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]).view(1, 1, 3, 3)
    ky = torch.tensor([[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]]).view(1, 1, 3, 3)
    ex = F.conv2d(x, kx, padding=1)
    ey = F.conv2d(x, ky, padding=1)
    edge = torch.sqrt(ex ** 2 + ey ** 2)
    edge = edge / edge.max()
    edge[edge >= high] = 1.0
    edge[edge <= low] = 0.0
    return Image.fromarray(edge[0, 0].mul(255).byte().numpy(), mode="L")


def reset_rng(seed: int):
    """videopipeline.py:110-112,126: the global CPU generator ends up in the state of a FRESH
    torch.Generator('cpu') on every frame, independent of `seed`."""
    np.random.seed(seed)
    torch.manual_seed(seed).set_state(torch.Generator(device="cpu").get_state())


def preprocess_image(img: Image.Image) -> torch.Tensor:
    """VaeImageProcessor.preprocess for a PIL RGB image whose size is already a multiple of 8."""
    a = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1)[None] * 2.0 - 1.0


def preprocess_control(img: Image.Image) -> torch.Tensor:
    """control_image_processor.preprocess: convert RGB, /255, NO normalisation (lcm_controlnet.py:109-113)."""
    a = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1)[None]


def postprocess(x: torch.Tensor) -> np.ndarray:
    """image_processor.postprocess(..., 'pil') up to the uint8 HWC array (lcm_controlnet.py:609-611)."""
    x = (x / 2 + 0.5).clamp(0, 1)
    a = x[0].permute(1, 2, 0).float().numpy()
    return (a * 255).round().astype("uint8")


class OraclePipeline:
    def __init__(self, unet_cfg, cn_cfg, w_unet, w_cn, w_vae, clip_cfg=None, w_clip=None, guidance_scale=7.5):
        self.unet_cfg, self.cn_cfg = unet_cfg, cn_cfg
        # fp32 copies of the (fp16-rounded) parameters, made ONCE: nets.py up-casts every weight where it uses it, which for
        # fp16 dictionaries was a third of a full-size frame's oracle time (3.4 GB of conversions per UNet pass)
        f32 = lambda w: None if w is None else {k: (v.float() if v.is_floating_point() else v) for k, v in w.items()}  # noqa: E731
        self.w_unet, self.w_cn, self.w_vae, self.w_clip = f32(w_unet), f32(w_cn), f32(w_vae), f32(w_clip)
        self.clip_cfg = clip_cfg
        self.sched = LCMSchedulerOracle()
        self.guidance_scale = guidance_scale  # never forwarded by videopipeline.py:114-124 -> always 7.5
        self.trace = {}

    def encode_prompt(self, ids: torch.Tensor) -> torch.Tensor:
        return nets.clip_text_forward(self.w_clip, self.clip_cfg, ids)

    @torch.no_grad()
    def infer(self, img: Image.Image, prompt_embeds: torch.Tensor, height=360, width=640, strength=0.4, steps=20,
              seed=42, controlnet_scale=1.0, use_controlnet=True, keep_trace=False, pooled=None,
              time_ids=None, ref_image: Image.Image = None, emulate_fp16: bool = False) -> Image.Image:
        """ref_image: run the reference-only mode of lcm_reference_pipeline.py:855-890 -- no ControlNet (that pipeline
        has none); per step a fresh noise draw for the reference latents (:861-871, taken from the same per-frame-reset
        CPU stream as every other draw of the frame, in call order), a WRITE pass WITHOUT the guidance embedding
        (:875-881 passes no timestep_cond) and the READ pass (:884-891).  The reference latents are the TAESD encoding of
        the reference image (`prepare_ref_latents` :161-209 calls `.latent_dist.sample`, which AutoencoderTiny does not
        have -- one reason this path is dead at v2; the live path's `retrieve_latents` behaviour is used)."""
        if emulate_fp16:  # (nets.EMULATE_FP16: every layer output rounded to fp16 -- how far fp16 STORAGE alone moves the result)
            nets.EMULATE_FP16 = True
            try:
                return self.infer(img, prompt_embeds, height=height, width=width, strength=strength, steps=steps, seed=seed,
                                  controlnet_scale=controlnet_scale, use_controlnet=use_controlnet, keep_trace=keep_trace, pooled=pooled,
                                  time_ids=time_ids, ref_image=ref_image)
            finally:
                nets.EMULATE_FP16 = False
        img = center_crop_resize(img, width, height)
        canny = sobel_edges(img, 0.11, 0.8)
        reset_rng(seed)
        image = preprocess_image(img)
        control = preprocess_control(canny)
        ts = self.sched.set_timesteps(strength, steps, 50)
        # prepare_latents (lcm_controlnet.py:250-337): TAESD encode, scaling_factor 1.0, global-RNG noise
        init = nets.taesd_encode(self.w_vae, image)
        noise = torch.randn(init.shape, dtype=init.dtype)
        latents = self.sched.add_noise(init, noise, ts[:1])
        w_emb = None
        if self.unet_cfg.cond_proj_dim:
            w_emb = w_embedding(torch.tensor(self.guidance_scale).repeat(1), self.unet_cfg.cond_proj_dim)
        text = prompt_embeds.float()
        added = None
        if self.unet_cfg.add_time_dim:  # SDXL: pooled text embeds + (orig size, crop, target size) micro-conditioning
            if time_ids is None:
                time_ids = torch.tensor([[height, width, 0, 0, height, width]], dtype=torch.float32)
            added = (pooled.float().reshape(1, -1), time_ids)
        if keep_trace:
            self.trace = {"init_latents": init.clone(), "noisy_latents": latents.clone(), "eps": [], "denoised": []}
        denoised = None
        ref_state = ref_latents = None
        if ref_image is not None:
            use_controlnet = False
            ref_latents = nets.taesd_encode(self.w_vae, preprocess_image(center_crop_resize(ref_image, width, height)))
            nlev = len(self.unet_cfg.block_out_channels)
            ref_state = nets.RefState(nlev, nlev)
            nets.rank_attention(self.unet_cfg, ref_state)
        for i, t in enumerate(ts):
            tt = torch.full((1,), int(t), dtype=torch.long)
            down = mid = None
            if ref_state is not None:
                ref_noise = torch.randn(ref_latents.shape, dtype=ref_latents.dtype)
                ref_xt = self.sched.add_noise(ref_latents, ref_noise, tt)
                ref_state.mode = "write"
                nets.unet_forward(self.w_unet, self.unet_cfg, ref_xt, tt, text, None, None, None, added, ref=ref_state)
                ref_state.mode = "read"
            if use_controlnet:
                down, mid = nets.controlnet_forward(self.w_cn, self.cn_cfg, latents, tt, text, control,
                                                    conditioning_scale=controlnet_scale, guess_mode=True)
            eps = nets.unet_forward(self.w_unet, self.unet_cfg, latents, tt, text, w_emb, down, mid, added, ref=ref_state)
            latents, denoised = self.sched.step(eps, i, t, latents)
            if keep_trace:
                self.trace["eps"].append(eps.clone())
                self.trace["denoised"].append(denoised.clone())
        out = nets.taesd_decode(self.w_vae, denoised)
        if keep_trace:
            self.trace["decoded"] = out.clone()
        return Image.fromarray(postprocess(out), mode="RGB")
