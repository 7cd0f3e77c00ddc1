"""What tests/golden/fullsize_oracle.npz was computed FROM: a sha256 over the oracle's sources, the weight generator, the
network configurations and the case parameters below.  scripts/make_fullsize_golden.py stores the digest in the fixture;
tests/test_oracle_golden.py fails when the stored digest no longer matches these files ("regenerate"), so that an edit to
oracle/nets.py or weights.synthesize cannot leave the full-size parity tests comparing against a stale answer."""
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_FULLSIZE = os.path.join(ROOT, "tests", "golden", "fullsize_oracle.npz")

# the inputs of the stored cases (scripts/make_fullsize_golden.py and the tests that read the fixture both take them from here)
CASES = {
    # BASELINE configs[4]: tests/test_pipeline_gpu.py::test_baseline_config5_768_eight_step_scale2_matches_oracle
    "config5": dict(nets="sd15", H=768, W=768, steps=8, strength=0.6, cn=True, cn_scale=2.0, frame_seed=41, text_seed=7, weights="cuda"),
    # ...::test_reference_only_mode_512_four_step_matches_oracle
    "ref512": dict(nets="sd15", H=512, W=512, steps=4, strength=0.6, cn=False, ref_seed=52, frame_seed=51, text_seed=7, weights="cuda"),
    # BASELINE configs[3]: tests/test_sdxl_gpu.py::test_sdxl_1024_four_step_matches_oracle
    "sdxl1024": dict(nets="sdxl", H=1024, W=1024, steps=4, strength=0.6, cn=False, frame_seed=2, text_seed=11, weights="cuda"),
    # BASELINE configs[1] on the RANGE-STRESS weight set (weights.synthesize(stress=True): per-channel scales over two decades, outlier
    # channels, norm gamma / beta away from (1, 0), attention logits to +-30): tests/test_pipeline_gpu.py::
    # test_baseline_config2_on_range_stress_weights_matches_oracle.  The UNet and the ControlNet are stressed, TAESD is not (it has no
    # normalisation layer: its trained weights are what keeps its activations in range, a random rescale is not a checkpoint)
    "stress512": dict(nets="sd15", H=512, W=512, steps=4, strength=0.6, cn=True, cn_scale=1.0, frame_seed=71, text_seed=7, weights="cuda",
                      stress=2),
    # ... and the milder level (half the decades, no pushed-out attention logits), where fp16 storage moves the result far less
    "stress512m": dict(nets="sd15", H=512, W=512, steps=4, strength=0.6, cn=True, cn_scale=1.0, frame_seed=71, text_seed=7, weights="cuda",
                       stress=1),
    # the small case the GPU suite ALSO computes live, so that the stored and the live comparison cannot drift apart
    # (tests/test_pipeline_gpu.py::test_stored_and_live_oracle_comparisons_agree); CPU-generator weights: reproducible anywhere
    "mini64": dict(nets="mini", H=64, W=64, steps=2, strength=0.6, cn=True, cn_scale=1.5, frame_seed=61, text_seed=7, weights="cpu"),
}

GUARDED = ["oracle/*.py", "videosd_amd/weights.py", "videosd_amd/config.py"]


def guarded_files():
    out = []
    for pat in GUARDED:
        out += sorted(glob.glob(os.path.join(ROOT, pat)))
    return [os.path.relpath(p, ROOT) for p in out]


def digests():
    """-> (overall sha256 hex, {relative path or 'cases': sha256 hex})"""
    per = {}
    for rel in guarded_files():
        with open(os.path.join(ROOT, rel), "rb") as f:
            per[rel] = hashlib.sha256(f.read()).hexdigest()
    per["cases"] = hashlib.sha256(json.dumps(CASES, sort_keys=True).encode()).hexdigest()
    total = hashlib.sha256(json.dumps(per, sort_keys=True).encode()).hexdigest()
    return total, per
