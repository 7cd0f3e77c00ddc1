"""A prepared frame program exported to a file and run by the C entry points alone (include/vsd.h vsd_plan_load / vsd_plan_infer):
what a host without Python gets instead of SURVEY.md section 8b's whole-frame calls.  The frame a plan produces is bit for bit the
Python engine's."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(batch=1, cn=True, H=128, W=96, steps=2):
    from videosd_amd import config as C, weights as W_
    from videosd_amd.engine import Engine
    from videosd_amd.ops import HipOps

    wu = W_.synthesize(W_.unet_spec(C.MINI_UNET), "unet.", device="cuda")
    wc = W_.synthesize(W_.controlnet_spec(C.MINI_CONTROLNET), "cn.", device="cuda")
    wv = W_.synthesize(W_.taesd_spec(C.TAESD), "vae.", device="cuda")
    eng = Engine(HipOps(0), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds((torch.randn(77, C.MINI_UNET.cross_dim, generator=torch.Generator().manual_seed(3)) * 0.5).half())
    eng.prepare(H, W, steps, 0.6, controlnet_scale=1.5, use_controlnet=cn, batch=batch)
    return eng


@pytest.mark.parametrize("batch,cn", [(1, True), (3, True), (1, False)])
def test_a_plan_file_run_through_the_c_entry_points_gives_the_engines_bits(tmp_path, batch, cn):
    from videosd_amd.plan import CPlan, export_plan

    H, W = 128, 96
    eng = _engine(batch=batch, cn=cn, H=H, W=W)
    rng = np.random.default_rng(5)
    shape = (H, W, 3) if batch == 1 else (batch, H, W, 3)
    frames = [rng.integers(0, 256, shape, dtype=np.uint8) for _ in range(3)]
    want = [eng.infer_u8(f).copy() for f in frames]
    path = str(tmp_path / "frame.vsdplan")
    info = export_plan(eng, path)
    assert info["calls"] > 100 and info["saved_bytes"] > 0 and os.path.getsize(path) > info["saved_bytes"]
    # the export ran the program once more on the engine's buffers: the engine itself is as it was
    assert np.array_equal(eng.infer_u8(frames[0]), want[0])
    plan = CPlan(path)
    try:
        assert (plan.H, plan.W, plan.batch) == (H, W, batch)
        for f, w in zip(frames, want):
            assert np.array_equal(plan.infer(f), w)
        assert np.array_equal(plan.infer(frames[0]), want[0])  # (nothing left over from the frame before)
        with pytest.raises(ValueError):
            plan.infer(np.zeros((8, 8, 3), np.uint8))
    finally:
        plan.close()


def test_a_file_that_is_not_a_plan_is_refused_by_name(tmp_path):
    from videosd_amd import lib as L
    import ctypes as C

    p = tmp_path / "junk.bin"
    p.write_bytes(b"not a plan at all" * 10)
    ctx = L.Context(0)
    h = C.c_void_p()
    with pytest.raises(RuntimeError, match="not a plan file"):
        ctx.call("vsd_plan_load", str(p).encode(), C.byref(h))
    with pytest.raises(RuntimeError, match="cannot open"):
        ctx.call("vsd_plan_load", str(tmp_path / "missing").encode(), C.byref(h))


def test_a_c_program_without_python_runs_the_plan(tmp_path):
    """examples/plan_host.c: gcc, libvsd.so, a plan file -- no Python in the process that denoises."""
    import subprocess

    from videosd_amd.plan import export_plan

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "plan_host")
    libdir = os.path.join(root, "videosd_amd")
    subprocess.run(["gcc", "-O2", os.path.join(root, "examples", "plan_host.c"), "-I" + os.path.join(root, "include"), "-L" + libdir, "-lvsd",
                    "-Wl,-rpath," + libdir, "-o", exe], check=True)
    eng = _engine(batch=2)
    frame = np.random.default_rng(9).integers(0, 256, (2, 128, 96, 3), dtype=np.uint8)
    want = eng.infer_u8(frame).copy()
    plan = str(tmp_path / "p.vsdplan")
    export_plan(eng, plan)
    (tmp_path / "in.raw").write_bytes(frame.tobytes())
    r = subprocess.run([exe, plan, str(tmp_path / "in.raw"), str(tmp_path / "out.raw"), "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    assert "frames/s" in r.stdout
    got = np.frombuffer((tmp_path / "out.raw").read_bytes(), dtype=np.uint8).reshape(frame.shape)
    assert np.array_equal(got, want)


def test_the_drop_in_class_exports_the_program_of_an_infer_call(tmp_path):
    """VideoSDPipeline.export_plan(path, **options): the frame `infer` returns for those options, from the C entry points."""
    from PIL import Image

    from videosd_amd.pipeline import VideoSDPipeline
    from videosd_amd.plan import CPlan

    pipe = VideoSDPipeline(model="SimianLuo/LCM_Dreamshaper_v7", controlnet="lllyasviel/control_v11p_sd15_canny", gpus=1, compile=False)
    opts = dict(prompt="a watercolor painting", height=192, width=256, strength=0.6, steps=2, seed=7, controlnet_scale=1.5)
    rng = np.random.default_rng(2)
    img = Image.fromarray(rng.integers(0, 256, (192, 256, 3), dtype=np.uint8), "RGB")  # (already the target size: no resampling on the way)
    want = np.asarray(pipe.infer(img, **opts))
    path = str(tmp_path / "infer.vsdplan")
    info = pipe.export_plan(path, **opts)
    assert info["calls"] > 500
    # another prompt for the same plan: its constant block as a file
    other = dict(opts, prompt="a charcoal sketch of a harbour")
    want_other = np.asarray(pipe.infer(img, **other))
    assert not np.array_equal(want_other, want)
    ppath = str(tmp_path / "other.vsdprompt")
    assert pipe.export_prompt(ppath, other["prompt"]) > 1 << 20
    plan = CPlan(path)
    try:
        assert (plan.H, plan.W, plan.batch) == (192, 256, 1)
        assert np.array_equal(plan.infer(np.asarray(img)), want)
        plan.load_prompt(ppath)
        assert np.array_equal(plan.infer(np.asarray(img)), want_other)
        with pytest.raises(RuntimeError, match="not a prompt file"):
            plan.load_prompt(path)
    finally:
        plan.close()


def test_truncated_and_corrupted_plan_files_are_refused_not_crashed_on(tmp_path):
    """Every prefix cut of a plan file, and a file with a pointer moved outside its allocation, ends in an error message."""
    import ctypes as C

    from videosd_amd import lib as L
    from videosd_amd.plan import export_plan

    eng = _engine(batch=1, cn=False, H=64, W=64, steps=1)
    eng.infer_u8(np.zeros((64, 64, 3), np.uint8))
    path = str(tmp_path / "ok.vsdplan")
    export_plan(eng, path)
    data = open(path, "rb").read()
    ctx = L.Context(0)
    h = C.c_void_p()
    for cut in (4, 30, 60, 200, 5000, len(data) // 3, len(data) - 1000):
        bad = tmp_path / f"cut{cut}.vsdplan"
        bad.write_bytes(data[:cut])
        with pytest.raises(RuntimeError, match="plan_load"):
            ctx.call("vsd_plan_load", str(bad).encode(), C.byref(h))
    # another interface version
    bad = tmp_path / "ver.vsdplan"
    bad.write_bytes(data[:8] + (99).to_bytes(4, "little") + data[12:])
    with pytest.raises(RuntimeError, match="bad header"):
        ctx.call("vsd_plan_load", str(bad).encode(), C.byref(h))
    # the frame buffer's offset past its region
    bad = tmp_path / "off.vsdplan"
    bad.write_bytes(data[:36] + (1 << 40).to_bytes(8, "little") + data[44:])
    with pytest.raises(RuntimeError, match="plan_load"):
        ctx.call("vsd_plan_load", str(bad).encode(), C.byref(h))
    # and the intact file still loads afterwards (nothing leaked into the context's state)
    ctx.call("vsd_plan_load", path.encode(), C.byref(h))
    ctx.lib.vsd_plan_free(ctx.h, h)
