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
    down_attn: Tuple[bool, ...] = (True, True, True, False)
    layers_per_block: int = 2
    heads: int = 8
    cross_dim: int = 768
    groups: int = 32
    cond_proj_dim: Optional[int] = 256  # LCM guidance embedding (time_cond_proj_dim); None for ControlNet
    text_len: int = 77

    @property
    def temb_dim(self) -> int:
        return 4 * self.block_out_channels[0]

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
SD15_CONTROLNET = ControlNetConfig()
TAESD = TAESDConfig()
CLIP_L = CLIPTextConfig()

# Reduced-width configuration with the same topology; every kernel path (concat gather, stride-2,
# upsample-to-size, head_dim padding, split-K) is exercised, and the CPU oracle finishes in seconds.
MINI_UNET = UNetConfig(block_out_channels=(64, 128, 256, 256), cross_dim=128, cond_proj_dim=64)
MINI_CONTROLNET = ControlNetConfig(unet=UNetConfig(block_out_channels=(64, 128, 256, 256), cross_dim=128,
                                                    cond_proj_dim=None), cond_channels=(16, 32, 96, 256))
MINI_CLIP = CLIPTextConfig(vocab=1000, width=128, heads=2, layers=2, mlp=512)
