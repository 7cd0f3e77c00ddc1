"""Pin oracle/nets.py to `diffusers` -- the day the dependency exists.

The reference's UNet / ControlNet / TAESD arithmetic lives in `diffusers` (unvendored: /root/reference/diffusert/requirements.txt:1;
call sites lcm_controlnet.py:299, 558-577, 594), which this image does not have, so `oracle/nets.py` is a restatement whose
parity is UNPINNED for those three forwards.  This script closes that wherever `import diffusers` succeeds
(a build container with the wheel; never the GPU box, never shipped): it builds diffusers' own UNet2DConditionModel /
ControlNetModel / AutoencoderTiny in the reduced MINI configuration AND the full SD1.5 configuration, loads the SAME seeded
synthetic state dicts the tests use (videosd_amd/weights.py names ARE diffusers' parameter names), runs both on the same seeded
inputs in fp32 on the CPU, compares, and writes inputs + diffusers' outputs as tests/golden/nets_<config>.npz.
tests/test_oracle_nets.py::test_oracle_nets_against_diffusers_goldens then holds oracle/nets.py to those files on every CPU run
(skipped while none exists).  No golden is written unless the two agree to 1e-4 relative L2: a disagreement is a finding to
read, not a fixture to commit.

    python scripts/pin_oracle_nets.py [--full]      # --full adds the 860 M-parameter SD1.5 configuration (~2 min of CPU)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

try:
    import diffusers  # noqa: F401
    from diffusers import AutoencoderTiny, ControlNetModel, UNet2DConditionModel
except Exception as e:  # the normal case in this image
    print(f"pin_oracle_nets: `import diffusers` failed ({type(e).__name__}: {e}); nothing pinned, nothing written")
    sys.exit(3)

from oracle import nets as N  # noqa: E402
from videosd_amd import config as C  # noqa: E402
from videosd_amd import weights as W  # noqa: E402


def unet_kwargs(cfg: C.UNetConfig) -> dict:
    """UNetConfig -> the constructor arguments of diffusers' UNet2DConditionModel / ControlNetModel for the same architecture"""
    nb = len(cfg.block_out_channels)
    down = tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in cfg.down_attn)
    kw = dict(in_channels=cfg.in_channels, block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block,
              down_block_types=down, cross_attention_dim=cfg.cross_dim, norm_num_groups=cfg.groups,
              attention_head_dim=cfg.heads if cfg.head_dim is None else tuple(c // cfg.head_dim for c in cfg.block_out_channels),
              use_linear_projection=cfg.linear_proj)
    if any(d > 1 for d in cfg.transformer_depth) or cfg.mid_depth != 1:
        kw["transformer_layers_per_block"] = tuple(max(1, d) for d in cfg.transformer_depth)
    assert nb == len(down)
    return kw


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def pin(tag: str, ucfg: C.UNetConfig, ccfg: C.ControlNetConfig, hw: int):
    g = torch.Generator().manual_seed(1234)
    f32 = lambda w: {k: v.float() for k, v in w.items()}  # noqa: E731
    wu = f32(W.synthesize(W.unet_spec(ucfg), "unet."))
    wc = f32(W.synthesize(W.controlnet_spec(ccfg), "cn."))
    wv = f32(W.synthesize(W.taesd_spec(C.TAESD), "vae."))
    up = tuple("CrossAttnUpBlock2D" if a else "UpBlock2D" for a in ucfg.up_attn)
    unet = UNet2DConditionModel(out_channels=ucfg.out_channels, up_block_types=up, time_cond_proj_dim=ucfg.cond_proj_dim, **unet_kwargs(ucfg)).eval()
    cnet = ControlNetModel(conditioning_embedding_out_channels=tuple(ccfg.cond_channels), conditioning_channels=ccfg.cond_in,
                           **unet_kwargs(ccfg.unet)).eval()
    vae = AutoencoderTiny().eval()
    for mod, w, what in ((unet, wu, "unet"), (cnet, wc, "controlnet"), (vae, wv, "taesd")):
        missing, unexpected = mod.load_state_dict(w, strict=False)
        assert not missing and not unexpected, f"{what}: diffusers' parameter names differ -- missing {missing[:4]}, unexpected {unexpected[:4]}"
    B = 1
    lat = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.tensor([499], dtype=torch.int64)
    text = torch.randn(B, ucfg.text_len, ucfg.cross_dim, generator=g) * 0.5
    wemb = torch.randn(B, ucfg.cond_proj_dim, generator=g) * 0.3 if ucfg.cond_proj_dim else None
    cond = torch.rand(B, 3, hw * 8, hw * 8, generator=g)
    img = torch.rand(B, 3, hw * 8, hw * 8, generator=g) * 2 - 1
    out = {}
    with torch.no_grad():
        d_down, d_mid = cnet(lat, t, encoder_hidden_states=text, controlnet_cond=cond, conditioning_scale=1.0, guess_mode=True, return_dict=False)
        o_down, o_mid = N.controlnet_forward(wc, ccfg, lat, t, text, cond, 1.0, True)
        errs = [rel(o, d) for o, d in zip(o_down + [o_mid], list(d_down) + [d_mid])]
        print(f"{tag} controlnet: max rel-L2 over 13 residuals {max(errs):.2e}")
        out.update({f"cn_down_{i}": d.numpy() for i, d in enumerate(d_down)}, cn_mid=d_mid.numpy())
        d_eps = unet(lat, t, encoder_hidden_states=text, timestep_cond=wemb, down_block_additional_residuals=d_down,
                     mid_block_additional_residual=d_mid, return_dict=False)[0]
        o_eps = N.unet_forward(wu, ucfg, lat, t, text, wemb, list(d_down), d_mid)
        errs.append(rel(o_eps, d_eps))
        print(f"{tag} unet: rel-L2 {errs[-1]:.2e}")
        out["unet_eps"] = d_eps.numpy()
        d_z = vae.encode(img).latents
        d_x = vae.decode(d_z).sample
        errs += [rel(N.taesd_encode(wv, img), d_z), rel(N.taesd_decode(wv, d_z), d_x)]
        print(f"{tag} taesd: encode {errs[-2]:.2e}, decode {errs[-1]:.2e}")
        out.update(taesd_z=d_z.numpy(), taesd_x=d_x.numpy())
    if max(errs) > 1e-4:
        print(f"{tag}: oracle/nets.py and diffusers {diffusers.__version__} DISAGREE (max rel-L2 {max(errs):.2e}): no golden written")
        return False
    path = os.path.join(ROOT, "tests", "golden", f"nets_{tag}.npz")
    np.savez_compressed(path, lat=lat.numpy(), t=t.numpy(), text=text.numpy(), wemb=np.zeros(0) if wemb is None else wemb.numpy(),
                        cond=cond.numpy(), img=img.numpy(), diffusers_version=np.array(diffusers.__version__), **out)
    print(f"{tag}: pinned against diffusers {diffusers.__version__} -> {path}")
    return True


if __name__ == "__main__":
    ok = pin("mini", C.MINI_UNET, C.MINI_CONTROLNET, 8)
    if "--full" in sys.argv:
        ok = pin("sd15", C.SD15_UNET, C.SD15_CONTROLNET, 16) and ok
    sys.exit(0 if ok else 1)
