"""The REAL-checkpoint path of the drop-in class (reference: videopipeline.py:49-72 loads SimianLuo/LCM_Dreamshaper_v7,
control_v11p_sd15_canny, madebyollin/taesd through diffusers; lcm_controlnet.py:115-198 tokenises and runs CLIP-L per frame):
safetensors files with diffusers / transformers key names under $VSD_WEIGHTS, tokenizer files beside them.  No checkpoint
exists offline, so the files are WRITTEN here from the seeded synthetic weights (every tensor under the key name a real
checkpoint uses) plus a toy byte-level BPE vocabulary -- what is exercised is the loader, the key
mapping, the missing-tensor check and the tokenizer -> HIP CLIP -> engine chain, none of which any earlier test ran
(VERDICT r2, A1 / A5)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

# tuning_mode="table": shapes the shipped table lacks take the deterministic heuristic instead of a per-instance timing run, so
# that two pipeline instances build the same kernels and "same weights" can be checked bit for bit
CFG = dict(model="SimianLuo/LCM_Dreamshaper_v7", controlnet="lllyasviel/control_v11p_sd15_canny", tuning_mode="table")
OPTS = dict(prompt="pixar, cg", height=128, width=192, strength=0.6, steps=2, controlnet_scale=1.0)


def write_toy_clip_tokenizer(d):
    """vocab.json / merges.txt in the format of openai/clip-vit-large-patch14's tokenizer (byte-level BPE, '</w>' word ends,
    <|startoftext|> / <|endoftext|>), with a handful of merges instead of 48 894."""
    from tokenizers import pre_tokenizers

    chars = sorted(pre_tokenizers.ByteLevel.alphabet())  # the 256 printable stand-ins of the byte values
    vocab = chars + [c + "</w>" for c in chars]
    merges = [("p", "i"), ("pi", "x"), ("a", "r</w>"), ("pix", "ar</w>"), ("c", "g</w>"), ("t", "h"), ("th", "e</w>")]
    vocab += ["".join(m) for m in merges]
    vocab += ["<|startoftext|>", "<|endoftext|>"]
    json.dump({t: i for i, t in enumerate(vocab)}, open(os.path.join(d, "vocab.json"), "w"))
    with open(os.path.join(d, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    return len(vocab)


@pytest.fixture(scope="module")
def weights_dir(tmp_path_factory):
    from safetensors.torch import save_file

    from videosd_amd import config as C
    from videosd_amd import weights as W

    d = str(tmp_path_factory.mktemp("vsd_weights"))
    cpu = lambda w: {k: v.cpu().contiguous() for k, v in w.items()}  # noqa: E731
    save_file(cpu(W.synthesize(W.unet_spec(C.SD15_UNET), "unet.", device="cuda")), os.path.join(d, "unet.safetensors"))
    save_file(cpu(W.synthesize(W.controlnet_spec(C.SD15_CONTROLNET), "cn.", device="cuda")), os.path.join(d, "controlnet.safetensors"))
    save_file(cpu(W.synthesize(W.taesd_spec(C.TAESD), "vae.", device="cuda")), os.path.join(d, "taesd.safetensors"))
    save_file(cpu(W.synthesize(W.clip_spec(C.CLIP_L), "clip.", device="cuda")), os.path.join(d, "text_encoder.safetensors"))
    write_toy_clip_tokenizer(d)
    return d


def _photo(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) // 2 + ((xx * 3 + yy * 5 + 40 * seed) % 256).astype(np.uint8)[..., None] // 2
    return Image.fromarray(a.astype(np.uint8), "RGB")


def test_checkpoint_files_give_the_frame_of_the_same_weights_and_the_tokenizer_feeds_the_hip_clip(weights_dir, monkeypatch):
    from oracle import nets
    from videosd_amd import config as C
    from videosd_amd import weights as W
    from videosd_amd.pipeline import VideoSDPipeline

    img = _photo(300, 200, 3)
    monkeypatch.delenv("VSD_WEIGHTS", raising=False)
    synth = VideoSDPipeline(**CFG)
    assert synth.text_encoder is None
    monkeypatch.setenv("VSD_WEIGHTS", weights_dir)
    real = VideoSDPipeline(**CFG)
    # (iii) the prompt goes tokenizer -> token ids -> HIP CLIP-L, and matches the oracle's CLIP on the same ids
    assert real.text_encoder is not None and real.text_encoder.has_tokenizer
    ids = real.text_encoder.tokenizer(OPTS["prompt"], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids[0]
    assert ids.shape == (77,) and ids[0] != ids[1] and int(ids.max()) < C.CLIP_L.vocab   # <|startoftext|>, tokens, <|endoftext|> padding
    assert 3 <= int((ids != ids[-1]).sum()) <= 12                                         # "pixar , cg" is a few BPE tokens, not 77
    emb = real.encode_prompt(OPTS["prompt"])
    wclip = {k: v.cpu() for k, v in W.load_safetensors(os.path.join(weights_dir, "text_encoder.safetensors")).items()}
    ref = nets.clip_text_forward(wclip, C.CLIP_L, ids[None])[0]
    got = emb.float().cpu()
    assert got.shape == (77, 768)
    assert float((got - ref).norm() / ref.norm()) < 5e-3
    # (i) same tensors through the files as through the seeded generator: bit-identical frame (both with the CLIP embeddings)
    a = np.asarray(real.infer(img, **OPTS))
    synth._prompts.clear()
    synth.set_prompt_embeds(emb, key=OPTS["prompt"])
    b = np.asarray(synth.infer(img, **OPTS))
    assert np.array_equal(a, b)
    # ... and the embeddings matter: the same pipeline with its seeded stand-in embeddings gives another frame
    synth._prompts.clear()
    c = np.asarray(synth.infer(img, **OPTS))
    assert np.abs(c.astype(int) - a.astype(int)).mean() > 1.0


def test_a_checkpoint_with_a_missing_tensor_is_refused(weights_dir, monkeypatch, tmp_path):
    """(ii) videopipeline.py:22-26 re-raises the loader's KeyError; here a checkpoint lacking a tensor the architecture needs
    names it instead of running with garbage."""
    from safetensors.torch import load_file, save_file

    from videosd_amd.pipeline import VideoSDPipeline

    d = str(tmp_path)
    for f in os.listdir(weights_dir):
        os.symlink(os.path.join(weights_dir, f), os.path.join(d, f))
    os.remove(os.path.join(d, "taesd.safetensors"))
    w = load_file(os.path.join(weights_dir, "taesd.safetensors"))
    w.pop("decoder.layers.0.weight")
    save_file(w, os.path.join(d, "taesd.safetensors"))
    monkeypatch.setenv("VSD_WEIGHTS", d)
    with pytest.raises(KeyError, match="taesd.safetensors.*decoder.layers.0.weight"):
        VideoSDPipeline(**CFG)
