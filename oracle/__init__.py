"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU fp32 restatement of the reference's per-frame algorithm
(/root/reference/diffusert/videopipeline.py:75-128 and diffusert/lcm/lcm_controlnet.py:379-618,
713-1071, diffusert/lcm/canny_gpu.py:6-44).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product (`videosd_amd/`) never does and fails
loudly when its HIP library is missing.

Pinning status
  * scheduler / w-embedding / RNG contract: PINNED — checked against tests/golden/lcm_scheduler.json,
    which was produced by executing the reference's own in-tree code (tests/golden/make_golden.py).
  * CLIP text encoder: PINNED against `transformers.CLIPTextModel` (installed here) on random weights; the SDXL pair
    (text_encoders.py: penultimate states of both towers, `CLIPTextModelWithProjection.text_embeds`) likewise.
  * UNet / ControlNet / TAESD network forward passes: PARITY UNPINNED.  Their arithmetic lives in the
    unvendored, unpinned third-party dependency `diffusers` (reference requirements.txt:1, bracketed to
    0.23-0.25 by its API use) which is not installable here; the restatement follows that library's
    published architecture and is anchored only by parameter counts
    (859.60 M / 361.28 M / 1.22 M) and by the reference's call sites.
"""
