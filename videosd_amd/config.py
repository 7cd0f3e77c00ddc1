"""Architecture descriptions for the networks on the per-frame path.

The reference never states these itself: it loads `SimianLuo/LCM_Dreamshaper_v7` (UNet),
`lllyasviel/control_v11p_sd15_canny` (ControlNet) and `madebyollin/taesd` through diffusers
(/root/reference/diffusert/videopipeline.py:49-72).  The shapes below restate those models'
published configs; parameter counts are checked in tests/test_weights.py
(859.60 M / 361.28 M / 1.22 M / 123.06 M).
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    transformer_depth: Tuple[int, ...] = (1, 1, 1, 0)  # BasicTransformerBlocks per attention at each level (0: none)
    mid_depth: int = 1
    layers_per_block: int = 2
    heads: int = 8                      # SD1.5: `attention_head_dim=8` means 8 heads at every level
    head_dim: Optional[int] = None      # SDXL: fixed head size (64), heads = C // head_dim
    cross_dim: int = 768
    groups: int = 32
    cond_proj_dim: Optional[int] = 256  # LCM guidance embedding (time_cond_proj_dim); None for ControlNet
    text_len: int = 77
    linear_proj: bool = False           # use_linear_projection: proj_in / proj_out are Linear instead of 1x1 conv
    add_time_dim: Optional[int] = None  # SDXL addition_embed_type="text_time": sinusoid width per time id (256)
    add_pooled_dim: int = 1280          # pooled text embedding width entering add_embedding
    add_n_ids: int = 6                  # (orig_h, orig_w, crop_top, crop_left, target_h, target_w)

    @property
    def temb_dim(self) -> int:
        return 4 * self.block_out_channels[0]

    @property
    def down_attn(self) -> Tuple[bool, ...]:
        return tuple(d > 0 for d in self.transformer_depth)

    @property
    def up_depth(self) -> Tuple[int, ...]:
        return tuple(reversed(self.transformer_depth))

    @property
    def add_in_dim(self) -> int:
        return self.add_pooled_dim + self.add_n_ids * self.add_time_dim if self.add_time_dim else 0

    def heads_for(self, c: int) -> int:
        return c // self.head_dim if self.head_dim else self.heads

    @property
    def up_attn(self) -> Tuple[bool, ...]:
        return tuple(reversed(self.down_attn))


@dataclass(frozen=True)
class ControlNetConfig:
    unet: UNetConfig = field(default_factory=lambda: UNetConfig(cond_proj_dim=None))
    cond_channels: Tuple[int, ...] = (16, 32, 96, 256)
    cond_in: int = 3


@dataclass(frozen=True)
class TAESDConfig:
    channels: int = 64
    latent_channels: int = 4
    image_channels: int = 3


@dataclass(frozen=True)
class CLIPTextConfig:
    vocab: int = 49408
    width: int = 768
    heads: int = 12
    layers: int = 12
    mlp: int = 3072
    max_len: int = 77
    eps: float = 1e-5


SD15_UNET = UNetConfig()
# BASELINE.json configs[3]: SDXL-base UNet with the LCM guidance projection (latent-consistency/lcm-sdxl), 2 567.55 M
# parameters.  Not in the reference: same per-frame loop, no ControlNet, TAESD-XL has TAESD's shape.
SDXL_UNET = UNetConfig(block_out_channels=(320, 640, 1280), transformer_depth=(0, 2, 10), mid_depth=10, head_dim=64,
                       cross_dim=2048, linear_proj=True, add_time_dim=256)
SD15_CONTROLNET = ControlNetConfig()
TAESD = TAESDConfig()
CLIP_L = CLIPTextConfig()

# Reduced-width configuration with the same topology; every kernel path (concat gather, stride-2,
# upsample-to-size, head_dim padding, split-K) is exercised, and the CPU oracle finishes in seconds.
MINI_UNET = UNetConfig(block_out_channels=(64, 128, 256, 256), cross_dim=128, cond_proj_dim=64)
MINI_CONTROLNET = ControlNetConfig(unet=UNetConfig(block_out_channels=(64, 128, 256, 256), cross_dim=128,
                                                    cond_proj_dim=None), cond_channels=(16, 32, 96, 256))
MINI_SDXL_UNET = UNetConfig(block_out_channels=(64, 128, 256), transformer_depth=(0, 2, 3), mid_depth=3, head_dim=32,
                            cross_dim=128, cond_proj_dim=64, linear_proj=True, add_time_dim=32, add_pooled_dim=96)
MINI_CLIP = CLIPTextConfig(vocab=1000, width=128, heads=2, layers=2, mlp=512)
