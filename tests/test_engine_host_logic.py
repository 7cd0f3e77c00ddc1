"""Host-logic test (no GPU): the engine's orchestration — weight packing, buffer wiring, skip/concat and
ControlNet-residual bookkeeping, schedule constants, non-divisible frame sizes — evaluated through the
TEST-ONLY FakeOps backend and compared with the oracle.  This checks the Python above the C-ABI, not the
kernels (tests/test_ops_gpu.py and tests/test_pipeline_gpu.py do that on the MI355X)."""
import numpy as np
import pytest
import torch
from PIL import Image

from fake_ops import FakeOps
from oracle.pipeline import OraclePipeline
from videosd_amd import config as C
from videosd_amd import weights as W
from videosd_amd.engine import Engine


@pytest.fixture(scope="module")
def mini():
    wu = W.synthesize(W.unet_spec(C.MINI_UNET), "unet.")
    wc = W.synthesize(W.controlnet_spec(C.MINI_CONTROLNET), "cn.")
    wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.")
    text = (torch.randn(77, C.MINI_UNET.cross_dim, generator=torch.Generator().manual_seed(7)) * 0.5).half()
    return wu, wc, wv, text


def _frame(h, w, seed=1):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = ((xx * 5 + yy * 3) % 256).astype(np.uint8)[..., None]
    return (base // 2 + grad // 2).astype(np.uint8)


@pytest.mark.parametrize("H,W,steps,cn", [(64, 64, 2, True), (48, 72, 1, True), (64, 64, 2, False), (56, 40, 2, True)])
def test_engine_program_matches_oracle(mini, H, W, steps, cn):
    wu, wc, wv, text = mini
    eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    plan = eng.prepare(H, W, steps, 0.6, controlnet_scale=1.5, use_controlnet=cn, use_graph=False)
    frame = _frame(H, W)
    got = eng.infer_u8(frame)
    orc = OraclePipeline(C.MINI_UNET, C.MINI_CONTROLNET, wu, wc, wv)
    ref = np.asarray(orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=H, width=W, strength=0.6,
                               steps=steps, seed=23, controlnet_scale=1.5, use_controlnet=cn, keep_trace=True))
    assert plan["timesteps"] == orc.sched.timesteps.tolist()
    # intermediate latents (fp16 storage vs fp32 oracle)
    h0, w0 = H // 8, W // 8
    x0 = eng.buffers["x0"][:, :4].float().reshape(h0, w0, 4).permute(2, 0, 1)
    ref_x0 = orc.trace["init_latents"][0]
    assert (x0 - ref_x0).norm() / ref_x0.norm() < 5e-3
    den = eng.buffers["denoised"][:, :4].float().reshape(h0, w0, 4).permute(2, 0, 1)
    ref_den = orc.trace["denoised"][-1][0]
    rel = float((den - ref_den).norm() / ref_den.norm())
    assert rel < 2e-2, rel
    diff = np.abs(got.astype(int) - ref.astype(int))
    assert diff.mean() < 1.5, diff.mean()


@pytest.mark.parametrize("H,Wd,steps", [(64, 64, 2), (56, 40, 1)])
def test_sdxl_shaped_engine_program_matches_oracle(H, Wd, steps):
    """BASELINE.json configs[3] topology (SDXL: 3 levels, no attention at level 0, several BasicTransformerBlocks per
    Transformer2D, Linear proj_in/out, fixed head size, text_time added conditioning, no ControlNet) at reduced width."""
    cfg = C.MINI_SDXL_UNET
    wu = W.synthesize(W.unet_spec(cfg), "xl.")
    wv = W.synthesize(W.taesd_spec(C.TAESD), "vae.")
    g = torch.Generator().manual_seed(11)
    text = (torch.randn(77, cfg.cross_dim, generator=g) * 0.5).half()
    pooled = (torch.randn(cfg.add_pooled_dim, generator=g) * 0.5).half()
    eng = Engine(FakeOps(), cfg, None, C.TAESD, wu, None, wv)
    eng.set_text_embeds(text)
    with pytest.raises(RuntimeError):
        eng.prepare(H, Wd, steps, 0.6, use_controlnet=False, use_graph=False)  # added conditioning missing
    eng.set_added_cond(pooled, (H, Wd, 0, 0, H, Wd))
    plan = eng.prepare(H, Wd, steps, 0.6, use_controlnet=False, use_graph=False)
    frame = _frame(H, Wd)
    got = eng.infer_u8(frame)
    orc = OraclePipeline(cfg, None, wu, None, wv)
    ref = np.asarray(orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=H, width=Wd, strength=0.6,
                               steps=steps, seed=23, use_controlnet=False, keep_trace=True, pooled=pooled))
    assert plan["timesteps"] == orc.sched.timesteps.tolist()
    h0, w0 = H // 8, Wd // 8
    den = eng.buffers["denoised"][:, :4].float().reshape(h0, w0, 4).permute(2, 0, 1)
    ref_den = orc.trace["denoised"][-1][0]
    rel = float((den - ref_den).norm() / ref_den.norm())
    assert rel < 2e-2, rel
    assert np.abs(got.astype(int) - ref.astype(int)).mean() < 1.5


@pytest.mark.parametrize("H,Wd,cn", [(64, 64, True), (56, 40, True), (48, 72, False)])
def test_batched_frames_equal_single_frames(mini, H, Wd, cn):
    """Several frames stacked along M in one program (prepare(batch=B)) give every frame the result it gets alone:
    per-image conv padding, GroupNorm statistics, self-attention keys, Sobel maximum and noise draws."""
    wu, wc, wv, text = mini
    frames = np.stack([_frame(H, Wd, seed=s) for s in (1, 2, 3)])
    eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    eng.prepare(H, Wd, 2, 0.6, controlnet_scale=1.5, use_controlnet=cn, use_graph=False)
    single = np.stack([eng.infer_u8(f) for f in frames])
    eng.prepare(H, Wd, 2, 0.6, controlnet_scale=1.5, use_controlnet=cn, use_graph=False, batch=3)
    assert eng.plan["batch"] == 3
    got = eng.infer_u8(frames)
    assert got.shape == single.shape
    diff = np.abs(got.astype(int) - single.astype(int))  # torch's batched conv sums in a different order: a few LSB flips
    assert diff.max() <= 4 and diff.mean() < 0.25, (diff.max(), diff.mean())
    with pytest.raises(ValueError):
        eng.infer_u8(frames[0])


def test_prepare_rejects_bad_sizes_and_missing_text(mini):
    wu, wc, wv, text = mini
    eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    with pytest.raises(RuntimeError):
        eng.prepare(64, 64, 2, 0.6, use_graph=False)
    eng.set_text_embeds(text)
    with pytest.raises(ValueError):
        eng.prepare(60, 64, 2, 0.6, use_graph=False)
    with pytest.raises(ValueError):
        eng.prepare(64, 64, 4, 0.01, use_graph=False)  # empty schedule (reference would crash at :588)


def test_update_options_equals_a_fresh_prepare(mini):
    """strength / controlnet_scale live in device constants the program READS (scheduler coefficients, ControlNet residual
    scales, per-step time embeddings): `update_options` on a prepared plan gives exactly what preparing with those options
    gives, slots follow their parent, and a strength that changes the NUMBER of timesteps is refused (the program itself
    would change)."""
    wu, wc, wv, text = mini
    frame = _frame(64, 64)

    def fresh(strength, scale):
        e = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
        e.set_text_embeds(text)
        e.prepare(64, 64, 2, strength, controlnet_scale=scale, use_controlnet=True, use_graph=False)
        return e.infer_u8(frame), e.plan["timesteps"]

    eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    eng.prepare(64, 64, 2, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=False)
    n_ops = len(eng.program.calls)
    slot = eng.make_slot()
    slot.prepare(64, 64, 2, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=False)
    base = eng.infer_u8(frame)
    for strength, scale in [(0.6, 2.5), (0.8, 0.3), (0.5, 1.0)]:
        assert eng.update_options(strength, scale) is True
        assert len(eng.program.calls) == n_ops  # same recorded program: nothing was rebuilt
        ref, ts = fresh(strength, scale)
        assert eng.plan["timesteps"] == ts
        got = eng.infer_u8(frame)
        assert np.array_equal(got, ref) and not np.array_equal(got, base)
        assert np.array_equal(slot.infer_u8(frame), ref)  # the slot reads the same constant block
    assert eng.update_options(0.03, 1.0) is False  # int(50 * 0.03) = 1 candidate timestep: a 1-step schedule, another program


def test_a_slot_owns_its_staging_buffers_and_events(mini):
    """Two launches in flight (a parent engine and its slot) must not share pinned host buffers or timing events: a slot
    made AFTER the parent's first frame used to inherit them (found on the GPU: `elapsed_time` on an event the other
    lane had re-recorded)."""
    wu, wc, wv, text = mini
    eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    eng.prepare(64, 64, 2, 0.6, use_controlnet=True, use_graph=False)
    a, b = _frame(64, 64, seed=1), _frame(64, 64, seed=2)
    out_a = eng.infer_u8(a)                      # the parent's staging exists now
    slot = eng.make_slot()
    slot.prepare(64, 64, 2, 0.6, use_controlnet=True, use_graph=False)
    eng.submit_u8(a)
    slot.submit_u8(b)
    assert slot._staging()[1] is not eng._staging()[1] and slot._staging()[2] is not eng._staging()[2]
    assert np.array_equal(eng.collect_u8(), out_a)
    assert not np.array_equal(slot.collect_u8(), out_a)


@pytest.mark.parametrize("H,W,steps", [(64, 64, 2), (64, 64, 1)])
def test_reference_only_mode_matches_oracle(mini, H, W, steps):
    """SURVEY 8f-4 (lcm_reference_pipeline.py:498-794, 855-890): per step a WRITE pass over the noised reference latents
    banks self-attention keys / values and block-output statistics; the READ pass attends over [x ; bank] and AdaINs the
    gated block outputs.  Host wiring through the op emulator against the oracle restatement."""
    wu, wc, wv, text = mini
    eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    eng.set_text_embeds(text)
    eng.prepare(H, W, steps, 0.6, use_controlnet=False, use_graph=False, ref_mode=True)
    frame, refimg = _frame(H, W, seed=1), _frame(H, W, seed=9)
    eng.ops.upload(eng.ref_u8, torch.from_numpy(refimg))
    got = eng.infer_u8(frame)
    orc = OraclePipeline(C.MINI_UNET, C.MINI_CONTROLNET, wu, wc, wv)
    ref = np.asarray(orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=H, width=W, strength=0.6, steps=steps,
                               seed=23, ref_image=Image.fromarray(refimg, "RGB"), keep_trace=True))
    h0, w0 = H // 8, W // 8
    den = eng.buffers["denoised"][:, :4].float().reshape(h0, w0, 4).permute(2, 0, 1)
    ref_den = orc.trace["denoised"][-1][0]
    rel = float((den - ref_den).norm() / ref_den.norm())
    assert rel < 2e-2, rel
    assert np.abs(got.astype(int) - ref.astype(int)).mean() < 1.5
    # and the mode does something: the plain UNet-only frame differs
    plain = np.asarray(orc.infer(Image.fromarray(frame, "RGB"), text[None].float(), height=H, width=W, strength=0.6, steps=steps,
                                 seed=23, use_controlnet=False))
    assert np.abs(plain.astype(int) - ref.astype(int)).mean() > 2.0
    with pytest.raises(ValueError):
        eng.prepare(H, W, steps, 0.6, use_controlnet=True, use_graph=False, ref_mode=True)


def test_absorbed_cross_attention_wiring(mini, monkeypatch):
    """Cross-attention of the wide blocks as two GEMMs (text K / V folded into the query / output weights, tile softmax
    in the first GEMM's epilogue; packing.pack_cross_attention): same frame as the three-kernel form, also after a prompt
    change (the folded weights are rewritten in place), through the op emulator with the width threshold lowered so that
    the reduced-width test network takes the path."""
    import videosd_amd.engine as E

    wu, wc, wv, text = mini
    frame = _frame(64, 64)
    monkeypatch.setattr(E, "XATTN_ABSORB_MIN_C", 1)
    eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
    assert all(t.xa_raw is not None for t in eng.unet.transformers)
    eng.set_text_embeds(text)
    eng.prepare(64, 64, 2, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=False)
    n_soft = sum(1 for fn, a, k in eng.flat_calls(eng.program.calls) if fn.__name__ == "conv" and k.get("softmax_cols"))
    n_attn = sum(1 for fn, a, k in eng.flat_calls(eng.program.calls) if fn.__name__ == "attention")
    nblk = len(eng.unet.transformers) + len(eng.cn.transformers)
    assert n_soft == 2 * nblk and n_attn == 2 * nblk  # every cross-attention absorbed, the self-attentions remain
    got = eng.infer_u8(frame)
    eng.absorb_cross_attention = False
    eng.prepare(64, 64, 2, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=False)
    assert sum(1 for fn, a, k in eng.flat_calls(eng.program.calls) if fn.__name__ == "attention") == 4 * nblk
    ref = eng.infer_u8(frame)
    assert np.abs(got.astype(int) - ref.astype(int)).mean() < 0.3
    # another prompt: the absorbed weights follow
    text2 = (torch.randn(77, C.MINI_UNET.cross_dim, generator=torch.Generator().manual_seed(8)) * 0.5).half()
    eng.set_text_embeds(text2)
    ref2 = eng.infer_u8(frame)
    eng.absorb_cross_attention = True
    eng.set_text_embeds(text2)
    eng.prepare(64, 64, 2, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=False)
    got2 = eng.infer_u8(frame)
    assert np.abs(got2.astype(int) - ref2.astype(int)).mean() < 0.3 and np.abs(ref2.astype(int) - ref.astype(int)).mean() > 1.0


def test_fused_tail_wiring(mini, monkeypatch):
    """The 320-wide transformer blocks run their per-token chains as tail_a / tail_b (csrc/fused_tail.hip): through the op
    emulator (its tail ops restate the chain in fp32) the frame equals the unfused program's, with 5 launches fewer per
    block.  The emulator's width constant is set to the reduced-width network's first level."""
    wu, wc, wv, text = mini
    frame = _frame(64, 64)
    c0 = C.MINI_UNET.block_out_channels[0]
    outs, counts = [], []
    for fused, min_rows in ((False, 0), (True, 1), (True, 1 << 30)):  # unfused / both chains fused / only the attention-side chain
        ops = FakeOps()
        ops.TAIL_C = c0
        eng = Engine(ops, C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
        eng.use_fused_tail = fused
        eng.tail_b_min_rows = min_rows
        eng.set_text_embeds(text)
        eng.prepare(64, 64, 2, 0.6, controlnet_scale=1.0, use_controlnet=True, use_graph=False)
        counts.append(len(eng.program.calls))
        names = [fn.__name__ for fn, a, k in eng.flat_calls(eng.program.calls)]
        assert ("tail_a" in names) == fused and ("tail_b" in names) == (fused and min_rows == 1)
        outs.append(eng.infer_u8(frame))
    assert counts[1] < counts[2] < counts[0]
    assert np.abs(outs[0].astype(int) - outs[2].astype(int)).mean() < 0.3
    assert counts[1] < counts[0]
    assert np.abs(outs[0].astype(int) - outs[1].astype(int)).mean() < 0.3


def test_arena_gives_oversized_tensors_their_own_chunk_and_rewinds_to_the_same_addresses():
    """Many frames per launch (or large frames) need buffers beyond the arena's usual chunk size: they get a chunk of their
    own instead of an error, and mark / rewind still hands out identical addresses on every denoising step."""
    from videosd_amd.engine import Arena

    a = Arena(FakeOps(), chunk_bytes=1 << 16)
    small = a.alloc(16, 64)
    m = a.mark()
    first = [a.alloc(64, 1024), a.alloc(8, 8), a.alloc(300, 512), a.alloc(16, 16)]  # 128 KB and 300 KB: both over 64 KB
    peak = a.peak
    a.rewind(m)
    again = [a.alloc(64, 1024), a.alloc(8, 8), a.alloc(300, 512), a.alloc(16, 16)]
    assert [t.data_ptr() for t in first] == [t.data_ptr() for t in again] and a.peak == peak
    ptrs = sorted((t.data_ptr(), t.numel() * t.element_size()) for t in first + [small])
    assert all(p0 + n0 <= p1 for (p0, n0), (p1, _) in zip(ptrs, ptrs[1:]))  # nothing overlaps


def _packed_convs(net):
    """every PackedConv object a NetWeights holds (any depth of its containers)"""
    from videosd_amd.packing import PackedConv

    out, seen, stack = [], set(), [net]
    while stack:
        o = stack.pop()
        if id(o) in seen:
            continue
        seen.add(id(o))
        if isinstance(o, PackedConv):
            out.append(o)
        elif isinstance(o, (list, tuple)):
            stack += list(o)
        elif isinstance(o, dict):
            stack += list(o.values())
        elif hasattr(o, "__dict__") and not isinstance(o, (torch.Tensor, type)):
            stack += list(vars(o).values())
    return out


def test_shortcut_convs_share_conv1s_grid_in_every_form_of_the_program(mini):
    """A ResnetBlock's shortcut conv depends on the block's input only (ResnetBlock2D.conv_shortcut): it is recorded as a member of
    conv1's grouped launch (ops.conv_group, every member at its own split), in the two-stream and in the lock-step form alike,
    one group per ResnetBlock that has a shortcut; VSD_NO_GROUP_SHORTCUT-style engines record it as a launch of its own and give
    the same frame."""
    wu, wc, wv, text = mini
    steps = 2

    def build(group):
        eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
        eng.set_text_embeds(text)
        eng.group_shortcuts = group
        eng.prepare(64, 64, steps, 0.6, use_controlnet=True, use_graph=False)
        return eng

    eng = build(True)
    n_sc = sum(1 for net in (eng.unet, eng.cn) for rw in _resnets(net) if rw.shortcut is not None)
    for prog in (eng.program, eng.program_serial):
        groups = [(a, k) for fn, a, k in prog.calls if fn.__name__ == "conv_group" and k.get("split") == "own"]
        assert len(groups) == n_sc * steps
        for a, k in groups:
            (a1, k1), (a2, k2) = a[0]
            assert a1[2].ksize == 3 and "rowvec" in k1 and a2[2].ksize == 1 and a1[3].n == a2[3].n  # conv1 + the 1x1 shortcut
        sc_w = {id(rw.shortcut) for net in (eng.unet, eng.cn) for rw in _resnets(net) if rw.shortcut is not None}
        assert not any(fn.__name__ == "conv" and id(a[3]) in sc_w for fn, a, k in prog.calls)
    frame = np.random.default_rng(6).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    assert np.array_equal(eng.infer_u8(frame), build(False).infer_u8(frame))


def _resnets(net):
    out, stack, seen = [], [net], set()
    while stack:
        o = stack.pop()
        if id(o) in seen:
            continue
        seen.add(id(o))
        if type(o).__name__ == "ResnetW":
            out.append(o)
        elif isinstance(o, (list, tuple)):
            stack += list(o)
        elif isinstance(o, dict):
            stack += list(o.values())
        elif hasattr(o, "__dict__") and not isinstance(o, (torch.Tensor, type)):
            stack += list(vars(o).values())
    return out


def test_captured_program_is_a_sequence_of_single_branch_graphs_and_event_edges(mini):
    """Engine._capture: every run of kernel calls on one stream is one graph, every fork / join / signal / wait an event edge
    (record on the producing stream, wait on the consuming one), in program order; without a second stream the frame is ONE
    graph.  (Two forked hipGraphs in flight serialise on the HIP runtime: include/vsd.h vsd_seq.)"""
    wu, wc, wv, text = mini
    steps = 2

    def build(overlap, side=False, cn=True, twin=False):
        eng = Engine(FakeOps(), C.MINI_UNET, C.MINI_CONTROLNET, C.TAESD, wu, wc, wv)
        eng.set_text_embeds(text)
        eng.overlap_controlnet, eng.use_side_stream, eng.twin_encoders = overlap, side, twin
        eng.prepare(64, 64, steps, 0.6, use_controlnet=cn, use_graph=False)
        seq = eng._capture(eng.program)
        return eng, seq["items"]

    per_step = [("record", 0), ("wait", 1), ("graph", 1), ("graph", 0), ("record", 1), ("wait", 0)]
    eng, items = build(False)
    assert items == [("graph", 0)]
    eng, items = build(True, cn=False)
    assert items == [("graph", 0)]
    # the two encoders in lock step (round 5, the default): their twin calls are pairs on ONE stream -- one graph, no edges --
    # and a pair holds the UNet's and the ControlNet's call of one op, in that order
    eng, items = build(False, twin=True)
    assert items == [("graph", 0)]
    pairs = [a for fn, a, k in eng.program.calls if fn.__name__ == "pair"]
    assert len(pairs) > 20 * steps and all(a[0][0].__name__ == a[1][0].__name__ for a in pairs)
    convs = [a for a in pairs if a[0][0].__name__ == "conv"]
    unet_w = {id(w) for w in _packed_convs(eng.unet)}
    assert all(id(a[0][1][3]) in unet_w and id(a[1][1][3]) not in unet_w for a in convs)
    assert not any(fn.__name__ in Engine.SYNC_OPS for fn, a, k in eng.program.calls)
    # with both options on (the default) the engine holds BOTH forms of the program: two streams for a lone launch, lock step for a
    # launch among busy lanes -- the same calls on the same buffers, so either gives the frame the other gives
    eng, items = build(True, twin=True)
    assert [i[:2] for i in items] == [("graph", 0)] + (per_step + [("graph", 0)]) * steps
    assert eng.program_serial is not eng.program and eng._capture(eng.program_serial, serial=True)["items"] == [("graph", 0)]
    flat = lambda prog: sorted((fn.__name__, tuple(id(x) for x in a)) for fn, a, k in eng.flat_calls(prog.calls)  # noqa: E731
                               if fn.__name__ not in Engine.SYNC_OPS)
    assert flat(eng.program) == flat(eng.program_serial)
    frame = np.random.default_rng(5).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    outs = []
    for ov in (True, False):
        eng.overlap_launch = ov
        outs.append(eng.infer_u8(frame))
    assert np.array_equal(outs[0], outs[1])
    # the two encoders on two streams: per step  main | record(0) wait(1) | ControlNet graph on 1 | UNet encoder graph on 0 |
    # record(1) wait(0), then the merges + decoder (+ next step's head) on 0
    eng, items = build(True)
    kinds = [i[:2] for i in items]
    per_step = [("record", 0), ("wait", 1), ("graph", 1), ("graph", 0), ("record", 1), ("wait", 0)]
    assert kinds == [("graph", 0)] + (per_step + [("graph", 0)]) * steps
    waits = [i for i in items if i[0] == "wait"]
    recs = {i[2]: i[1] for i in items if i[0] == "record"}
    assert len(waits) == 2 * steps and all(recs[w[2]] != w[1] for w in waits)      # every wait is on the OTHER stream's event
    assert all(items.index(("record", recs[w[2]], w[2])) < items.index(w) for w in waits)
    assert eng.ops.seq_count({"items": items}) == (1 + 3 * steps, 2 * steps)
    # the side-stream option (ControlNet skip merges beside the decoder): more, smaller graphs; every named wait has its record
    eng, items = build(True, side=True)
    recs = {i[2] for i in items if i[0] == "record"}
    assert all(i[2] in recs for i in items if i[0] == "wait")
    assert sum(i[0] == "graph" for i in items) > 1 + 3 * steps
