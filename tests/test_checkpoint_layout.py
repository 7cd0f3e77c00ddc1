"""Where the drop-in class finds its checkpoints (CPU): the directories `from_pretrained` reads in the reference
(/root/reference/diffusert/videopipeline.py:51-69) -- a local directory or a hub id's snapshot in the local Hugging Face cache,
`<root>/<subfolder>/<stem>[.fp16].safetensors` -- then the flat $VSD_WEIGHTS files, then seeded synthetic tensors; what a file
must hold.  The GPU side (frames from such directories) is tests/test_checkpoint_gpu.py."""
import os

import pytest
import torch
from safetensors.torch import save_file

from videosd_amd import config as C
from videosd_amd import checkpoints as CK
from videosd_amd import weights as W
from videosd_amd.pipeline import load_or_synthesize


def test_find_snapshot_takes_a_directory_or_the_newest_revision_in_the_hf_cache(tmp_path, monkeypatch):
    d = tmp_path / "my_model"
    d.mkdir()
    assert CK.find_snapshot(str(d)) == str(d)
    cache = tmp_path / "hub"
    old = cache / "models--SimianLuo--LCM_Dreamshaper_v7" / "snapshots" / "aaaa"
    new = cache / "models--SimianLuo--LCM_Dreamshaper_v7" / "snapshots" / "bbbb"
    old.mkdir(parents=True)
    new.mkdir(parents=True)
    os.utime(old, (1, 1))
    monkeypatch.setenv("HF_HUB_CACHE", str(cache))
    assert CK.find_snapshot("SimianLuo/LCM_Dreamshaper_v7") == str(new)  # no refs/: the newest snapshot
    # ... but `from_pretrained` resolves refs/main: with a ref the OLDER revision it names wins over the newest directory
    refs = cache / "models--SimianLuo--LCM_Dreamshaper_v7" / "refs"
    refs.mkdir()
    (refs / "main").write_text("aaaa\n")
    assert CK.find_snapshot("SimianLuo/LCM_Dreamshaper_v7") == str(old)
    (refs / "main").write_text("cccc")  # a ref to a revision that is not cached: back to the newest
    assert CK.find_snapshot("SimianLuo/LCM_Dreamshaper_v7") == str(new)
    (refs / "pinned").write_text("aaaa")
    monkeypatch.setenv("VSD_HF_REVISION", "pinned")
    assert CK.find_snapshot("SimianLuo/LCM_Dreamshaper_v7") == str(old)
    monkeypatch.delenv("VSD_HF_REVISION")
    assert CK.find_snapshot("lllyasviel/control_v11p_sd15_canny") is None  # not cached: never downloaded, the caller falls back
    assert CK.find_snapshot(None) is None and CK.find_snapshot("") is None


def test_checkpoint_file_knows_the_names_diffusers_and_transformers_use(tmp_path):
    (tmp_path / "unet").mkdir()
    (tmp_path / "text_encoder").mkdir()
    assert CK.checkpoint_file(str(tmp_path), "unet") is None
    for rel in ("unet/diffusion_pytorch_model.fp16.safetensors", "text_encoder/model.safetensors", "diffusion_pytorch_model.safetensors"):
        (tmp_path / rel).write_bytes(b"")
    assert CK.checkpoint_file(str(tmp_path), "unet").endswith("unet/diffusion_pytorch_model.fp16.safetensors")
    (tmp_path / "unet" / "diffusion_pytorch_model.safetensors").write_bytes(b"")  # the plain name wins over the variant
    assert CK.checkpoint_file(str(tmp_path), "unet").endswith("unet/diffusion_pytorch_model.safetensors")
    assert CK.checkpoint_file(str(tmp_path), "text_encoder", stems=("model",)).endswith("text_encoder/model.safetensors")
    assert CK.checkpoint_file(str(tmp_path)).endswith("diffusion_pytorch_model.safetensors")
    (tmp_path / "vae").mkdir()
    (tmp_path / "vae" / "diffusion_pytorch_model.bin").write_bytes(b"")  # pickles are not loaded
    assert CK.checkpoint_file(str(tmp_path), "vae") is None and CK.checkpoint_file(None) is None


def test_load_order_snapshot_then_flat_then_synthetic_and_the_cast(tmp_path, monkeypatch):
    spec = W.taesd_spec(C.TAESD)
    synth = W.synthesize(spec, "vae.")
    monkeypatch.delenv("VSD_WEIGHTS", raising=False)
    w, src = load_or_synthesize(spec, "vae.", "taesd.safetensors", "cpu")
    assert src == "synthetic" and all(torch.equal(w[k], synth[k]) for k in synth)
    flat = tmp_path / "flat"
    flat.mkdir()
    other = {k: (v.float() * 2).contiguous() for k, v in synth.items()}  # fp32 on disk
    save_file(other, str(flat / "taesd.safetensors"))
    monkeypatch.setenv("VSD_WEIGHTS", str(flat))
    w, src = load_or_synthesize(spec, "vae.", "taesd.safetensors", "cpu")
    assert src.endswith("flat/taesd.safetensors") and all(v.dtype == torch.float16 for v in w.values())
    assert torch.equal(w["decoder.layers.0.weight"], (synth["decoder.layers.0.weight"].float() * 2).half())
    snap = tmp_path / "snap"
    snap.mkdir()
    third = {k: (v.float() * 3).half().contiguous() for k, v in synth.items()}
    third["some.buffer_of_ints"] = torch.arange(4)
    save_file(third, str(snap / "diffusion_pytorch_model.safetensors"))
    w, src = load_or_synthesize(spec, "vae.", "taesd.safetensors", "cpu", CK.checkpoint_file(str(snap)))
    assert src.startswith(str(snap)) and torch.equal(w["encoder.layers.0.bias"], third["encoder.layers.0.bias"])
    assert w["some.buffer_of_ints"].dtype == torch.int64  # integer buffers are not cast


def test_a_file_that_does_not_hold_the_architecture_is_refused_by_name(tmp_path):
    spec = W.taesd_spec(C.TAESD)
    w = W.synthesize(spec, "vae.")
    CK.check_against_spec(w, spec, "ok")
    bad = dict(w)
    bad.pop("encoder.layers.3.conv.2.weight")
    with pytest.raises(KeyError, match="taesd.*encoder.layers.3.conv.2.weight"):
        CK.check_against_spec(bad, spec, "taesd")
    bad = dict(w)
    bad["decoder.layers.0.weight"] = torch.zeros(64, 16, 3, 3)
    with pytest.raises(ValueError, match=r"decoder.layers.0.weight has shape \(64, 16, 3, 3\).*\(64, 4, 3, 3\)"):
        CK.check_against_spec(bad, spec, "taesd")
