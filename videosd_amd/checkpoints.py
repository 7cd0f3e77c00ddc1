"""Where the checkpoints are: the directories `from_pretrained` reads in the reference
(/root/reference/diffusert/videopipeline.py:51-69 -- ControlNetModel.from_pretrained(controlnet_model),
LatentConsistencyModelPipeline_controlnet.from_pretrained("SimianLuo/LCM_Dreamshaper_v7"), AutoencoderTiny.from_pretrained(
"madebyollin/taesd")), resolved offline, and what a file must hold (videosd_amd/weights.py has the architectures' tensor
lists).  Nothing here downloads anything."""
import os
from typing import Dict, Optional

import torch

from .weights import Spec


def load_safetensors(path: str, device="cpu", dtype=torch.float16) -> Dict[str, torch.Tensor]:
    """Every tensor of the file, cast like `from_pretrained(..., torch_dtype=torch.float16)` (fp32 checkpoints are rounded
    once to fp16; integer buffers such as CLIP's `position_ids` are left alone)."""
    from safetensors.torch import load_file

    return {k: (v.to(device=device, dtype=dtype) if v.is_floating_point() else v.to(device=device)) for k, v in load_file(path).items()}


# ------------------------------------------------------------------------------------------ `from_pretrained` layouts
def find_snapshot(name) -> Optional[str]:
    """The directory a `from_pretrained(name)` call of the reference would read (videopipeline.py:51-69), offline: `name` itself
    when it is a directory, else the snapshot `refs/main` names (the newest one when there is no ref) of that hub id in the local Hugging Face cache
    (`$HF_HUB_CACHE`, `$HF_HOME/hub`, `~/.cache/huggingface/hub`: `models--<org>--<repo>/snapshots/<rev>/`), else None
    (there is no network here: nothing is ever downloaded)."""
    if not name:
        return None
    name = str(name)
    if os.path.isdir(name):
        return name
    roots = [os.environ.get("HF_HUB_CACHE"), os.environ.get("HUGGINGFACE_HUB_CACHE"),
             os.path.join(os.environ["HF_HOME"], "hub") if os.environ.get("HF_HOME") else None,
             os.path.join(os.path.expanduser("~"), ".cache", "huggingface", "hub")]
    for root in roots:
        if not root:
            continue
        repo = os.path.join(root, "models--" + name.replace("/", "--"))
        snaps = os.path.join(repo, "snapshots")
        if os.path.isdir(snaps):
            # what `from_pretrained` resolves: the commit `refs/<revision>` names (main unless $VSD_HF_REVISION says otherwise) --
            # with several cached revisions the newest directory is not necessarily the one the reference would load
            ref = os.path.join(repo, "refs", os.environ.get("VSD_HF_REVISION", "main"))
            if os.path.isfile(ref):
                try:
                    rev = open(ref).read().strip()
                except OSError:
                    rev = ""
                if rev and os.path.isdir(os.path.join(snaps, rev)):
                    return os.path.join(snaps, rev)
            revs = [os.path.join(snaps, r) for r in os.listdir(snaps)]
            revs = [r for r in revs if os.path.isdir(r)]
            if revs:
                return max(revs, key=os.path.getmtime)  # (no usable ref: the newest snapshot)
    return None


def checkpoint_file(root: Optional[str], subfolder: str = "", stems=("diffusion_pytorch_model", "model")) -> Optional[str]:
    """`<root>/<subfolder>/<stem>[.fp16].safetensors` as diffusers / transformers name their single-file checkpoints
    (`unet/diffusion_pytorch_model.safetensors`, `text_encoder/model.safetensors`, a ControlNet's or TAESD's
    `diffusion_pytorch_model.safetensors` at the top level).  Only safetensors: `.bin` pickles and sharded checkpoints are
    refused by not being found."""
    if not root:
        return None
    d = os.path.join(root, subfolder) if subfolder else root
    for stem in stems:
        for variant in ("", ".fp16"):
            f = os.path.join(d, stem + variant + ".safetensors")
            if os.path.isfile(f):
                return f
    return None


def check_against_spec(w: Dict[str, torch.Tensor], spec: Spec, what: str):
    """A checkpoint must hold every tensor of the architecture, in its shape: name what is wrong instead of running on garbage
    (the reference lets `from_pretrained` raise; videopipeline.py:22-26 re-raises a KeyError)."""
    missing = [n for n, _, _ in spec if n not in w]
    if missing:
        raise KeyError(f"{what}: missing tensors {missing[:4]}{'...' if len(missing) > 4 else ''}")
    bad = [(n, tuple(w[n].shape), tuple(shape)) for n, shape, _ in spec if tuple(w[n].shape) != tuple(shape)]
    if bad:
        n, got, want = bad[0]
        raise ValueError(f"{what}: tensor {n} has shape {got}, the architecture needs {want} ({len(bad)} mismatching tensor(s))")
