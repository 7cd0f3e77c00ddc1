"""The oracle's scheduler / w-embedding / RNG contract against the golden vectors produced by the
reference's own code (tests/golden/make_golden.py).  Exact for integers, <= 1 ulp for fp32."""
import json
import os

import numpy as np
import torch

from oracle.scheduler import LCMSchedulerOracle, w_embedding
from oracle.pipeline import reset_rng

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lcm_scheduler.json")))


def _ulp_close(a, b, ulps=1):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return np.all(np.abs(a - b) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b))))


def test_timesteps_table():
    s = LCMSchedulerOracle()
    for c in G["timesteps"]:
        assert s.set_timesteps(c["strength"], c["steps"]).tolist() == c["timesteps"], c


def test_alphas_cumprod_bit_exact():
    s = LCMSchedulerOracle()
    ref = np.frombuffer(bytes.fromhex(G["alphas_cumprod_f32_hex"]), dtype=np.float32)
    assert np.array_equal(s.alphas_cumprod.numpy(), ref)


def test_scalings():
    for t, (cs, co) in G["scalings"].items():
        a, b = LCMSchedulerOracle.scalings(torch.tensor(int(t)))
        assert _ulp_close(float(a), cs) and _ulp_close(float(b), co)


def test_w_embedding():
    ref = np.frombuffer(bytes.fromhex(G["w_embedding_7p5_f32_hex"]), dtype=np.float32).reshape(1, 256)
    e = w_embedding(torch.tensor(7.5).repeat(1), 256).numpy()
    assert np.array_equal(e, ref)


def test_rng_contract():
    assert torch.Generator(device="cpu").initial_seed() == G["fresh_generator_initial_seed"]
    for seed, r in G["rng"].items():
        reset_rng(int(seed))
        assert torch.randn(4).tolist() == r["randn4"]
        reset_rng(int(seed))
        b = torch.randn(1, 4, 64, 64)
        c = torch.randn(1, 4, 64, 64)
        assert float(b.sum()) == r["draw0_sum"] and b[0, 0, 0, :4].tolist() == r["draw0_first4"]
        assert float(c.sum()) == r["draw1_sum"]


def test_step_traces():
    for rec in G["step_traces"]:
        s = LCMSchedulerOracle()
        ts = s.set_timesteps(rec["strength"], rec["steps"])
        x0 = torch.tensor(rec["sample0"]).view(1, 4, 8, 8)
        n0 = torch.tensor(rec["noise0"]).view(1, 4, 8, 8)
        lat = s.add_noise(x0, n0, ts[:1])
        assert _ulp_close(lat.flatten().numpy(), rec["noisy"])
        reset_rng(rec["steps"])
        for i, (t, it) in enumerate(zip(ts, rec["iters"])):
            assert int(t) == it["t"]
            eps = torch.tensor(it["eps"]).view(1, 4, 8, 8)
            prev, den = s.step(eps, i, t, lat)
            assert _ulp_close(den.flatten().numpy(), it["denoised"])
            assert _ulp_close(prev.flatten().numpy(), it["prev"], ulps=2)
            assert (prev is den) == it["prev_is_denoised"]
            lat = prev


def test_stored_fullsize_oracle_outputs_belong_to_the_current_oracle_sources():
    """tests/golden/fullsize_oracle.npz holds the oracle's OUTPUT for the slowest full-size cases; it is only a valid reference
    while the files it was computed from are the ones in the tree (nothing tied it to them).  The fixture carries the
    digest of oracle/*.py, videosd_amd/weights.py, videosd_amd/config.py and the case parameters (tests/golden_guard.py)."""
    import json

    import numpy as np

    import golden_guard as G

    with np.load(G.GOLDEN_FULLSIZE) as z:
        assert "guard_sha256" in z.files, "fixture without a source digest: regenerate with scripts/make_fullsize_golden.py"
        stored, stored_per = str(z["guard_sha256"]), json.loads(str(z["guard_files"]))
        for case in G.CASES:
            assert f"{case}_image_half" in z.files and f"{case}_denoised" in z.files, case
    total, per = G.digests()
    changed = sorted(k for k in set(per) | set(stored_per) if per.get(k) != stored_per.get(k))
    assert total == stored and not changed, (
        f"tests/golden/fullsize_oracle.npz was computed from other versions of {changed}: regenerate it on the GPU box with "
        "`python scripts/make_fullsize_golden.py` (or, when the edit cannot change the oracle's numbers, re-stamp it with --stamp "
        "and say why in the commit)")


def test_the_guard_notices_an_edited_oracle_source(tmp_path, monkeypatch):
    import shutil

    import golden_guard as G

    root = tmp_path / "tree"
    for rel in G.guarded_files():
        (root / rel).parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(G.ROOT + "/" + rel, root / rel)
    monkeypatch.setattr(G, "ROOT", str(root))
    before, _ = G.digests()
    assert before == G.digests()[0]
    with open(root / "oracle" / "nets.py", "a") as f:
        f.write("\n# an edit\n")
    after, per = G.digests()
    assert after != before
    monkeypatch.setitem(G.CASES["config5"], "cn_scale", 2.5)
    assert G.digests()[0] != after
