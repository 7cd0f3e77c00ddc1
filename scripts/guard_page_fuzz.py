"""Out-of-bounds READS and writes: random ragged shapes through the C-ABI ops with EVERY operand (inputs, weights, bias, residual,
statistics, split-K workspace, outputs) in a buffer that ends at -- or starts right after -- an unmapped page
(scripts/guard_pages.cpp: HIP virtual-memory API).  A kernel that touches one byte past an operand takes a GPU memory fault on
the spot, whatever the allocator would have put there; the case is printed BEFORE it runs, so the last line names it.  Results
are checked against fp32 references as well.
usage (GPU box): python scripts/guard_page_fuzz.py [seconds=90] [seed=0] [selftest]
(selftest: reads one element past a guarded buffer on purpose -- must die with a memory access fault)"""
import ctypes as C
import os, subprocess, sys, time
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videosd_amd import lib as L
from videosd_amd.ops import Geom, HipOps
from videosd_amd.packing import pack_conv, pack_cross_attention, pack_linear

so = os.path.join(ROOT, "scripts", "libguardpages.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "scripts", "guard_pages.cpp"), "-o", so])
ops = HipOps(0)  # (initialises the device before the helper's first call)
gp = C.CDLL(so)
gp.guard_alloc.restype = C.c_void_p
gp.guard_alloc.argtypes = [C.c_size_t, C.c_int]
gp.guard_free.argtypes = [C.c_void_p]
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TYPESTR = {torch.float16: "<f2", torch.float32: "<f4", torch.uint8: "|u1", torch.int32: "<i4", torch.int64: "<i8"}
live = []


class _Iface:
    def __init__(self, ptr, shape, dtype):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": TYPESTR[dtype], "data": (ptr, False), "version": 2}


def guarded(src: torch.Tensor = None, shape=None, dtype=torch.float16, fill=0.0):
    """a device tensor whose storage ends at an unmapped page (or, one time in four, starts right after one); `src`: its content"""
    if src is not None:
        src = src.contiguous()
        shape, dtype = tuple(src.shape), src.dtype
    n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
    ptr = gp.guard_alloc(max(n, 16), int(rng.random() < 0.75))
    assert ptr, "guard_alloc failed"
    live.append(ptr)
    t = torch.as_tensor(_Iface(ptr, shape, dtype), device="cuda")
    if src is not None:
        t.copy_(src)
    else:
        t.fill_(fill)
    return t


def release():
    # (nothing is unmapped while the run lasts: a virtual address that was unmapped and mapped again onto other pages read
    #  stale data in this harness -- the first version did that and "found" wrong results at random; a run maps a few GB)
    torch.cuda.synchronize()
    live.clear()


def rnd(*shape, scale=1.0):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * scale).half()


def close(got, ref, what, rel=4e-3):
    got, ref = got.float().cpu(), ref.float()
    assert torch.isfinite(got).all(), what
    l2 = float((got - ref).norm() / (ref.norm() + 1e-9))
    assert l2 <= rel, f"{what}: rel-L2 {l2:.3g}"


def gpack(pw):
    for f in ("weight", "bias", "ln_s", "ln_t", "weight_frag"):
        v = getattr(pw, f, None)
        if v is not None:
            setattr(pw, f, guarded(v))
    return pw


def one_conv():
    ks = int(rng.choice([1, 3]))
    stride = int(rng.choice([1, 1, 2])) if ks == 3 else 1
    h, w = int(rng.integers(1, 36)), int(rng.integers(1, 36))
    b = int(rng.choice([1, 1, 2, 3]))
    cin = int(rng.choice([64, 128, 192, 320, 640]))
    cout = int(rng.choice([8, 24, 64, 72, 136, 200, 320]))
    tile, pipeline = int(rng.choice([0, 1, 2, 3, 4, 5])), int(rng.choice([0, 3, 4, 5, 6, 7, 8, 10]))
    split, inkernel = int(rng.choice([1, 1, 2, 3, 5])), bool(rng.random() < 0.5)
    act, use_res = int(rng.choice([0, 1, 2, 4, 6])), bool(rng.random() < 0.5)
    if pipeline == 10:  # the persistent 64 -> 64 channel form (csrc/conv_c64.hip): the layer it exists for, on ragged images
        ks, stride, cin, cout, tile, split, act = 3, 1, 64, 64, 5, 1, int(rng.choice([0, 1, 2]))
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    up = ks == 3 and stride == 1 and rng.random() < (0.4 if pipeline == 10 else 0.2)
    g = Geom.conv(h, w, ksize=ks, stride=stride, batch=b, up_to=(2 * h, 2 * w) if up else None)
    split = min(split, (cin * ks * ks + 63) // 64)
    desc = f"conv {b}x{h}x{w} up {up} {cin}->{cout} ks{ks} s{stride} tile {tile} pipeline {pipeline} split {split} inkernel {inkernel} act {act} res {use_res}"
    print(desc, flush=True)
    x = rnd(b, cin, h, w)
    wt, bias = rnd(cout, cin, ks, ks, scale=(cin * ks * ks) ** -0.5), rnd(cout, scale=0.1)
    pw = gpack(pack_conv(wt, bias))
    src = guarded(x.permute(0, 2, 3, 1).reshape(-1, cin))
    ldo = (cout + 7) // 8 * 8
    out = guarded(shape=(g.m, ldo))
    ws = guarded(shape=(max(1, split) * g.m, cout), dtype=torch.float32) if split > 1 else None
    res = rnd(g.m, ldo) if use_res else None
    ops.inkernel_splitk = inkernel
    try:
        ops.conv(src, None, g, pw, out, ldo=ldo, act=act, residual=None if res is None else guarded(res), ldr=ldo,
                 tile=tile, split_k=split, pipeline=pipeline, workspace=ws)
        ops.synchronize()
    except RuntimeError:
        return "refused"
    finally:
        ops.inkernel_splitk = True
    xin = F.interpolate(x.float(), size=(2 * h, 2 * w), mode="nearest") if up else x.float()
    ref = F.conv2d(xin, wt.float(), bias.float(), stride=stride, padding=ks // 2)
    ref = {0: lambda v: v, 1: F.relu, 2: F.silu, 4: lambda v: v * torch.sigmoid(1.702 * v), 6: F.gelu}[act](ref)
    ref = ref.permute(0, 2, 3, 1).reshape(g.m, cout)
    if res is not None:
        ref = ref + res[:, :cout].float()
    close(out[:, :cout], ref, desc)
    return "ok"


def one_group():
    """2-4 independent convs / linear layers as ONE grid (vsd_conv_gemm_group): ragged members of different shapes, one kernel form,
    split over K with the slabs reduced in the launch or by the group's reducer -- every operand AND the members' split-K workspaces
    (HipOps.allocator hook) between unmapped pages; the result must be bit for bit what the members give launched one by one."""
    n = int(rng.integers(2, 5))
    tile, pipeline = int(rng.choice([0, 1, 2, 3])), int(rng.choice([3, 5]))
    split, inkernel = int(rng.choice([1, 1, 2, 3, 6])), bool(rng.random() < 0.5)
    members, refs, descs = [], [], []
    for i in range(n):
        ks = int(rng.choice([1, 1, 3]))
        h, w, b = int(rng.integers(1, 28)), int(rng.integers(1, 28)), int(rng.choice([1, 1, 2]))
        cin, cout = int(rng.choice([64, 128, 320])), int(rng.choice([8, 24, 64, 72, 136, 200, 320]))
        c1 = 64 if (cin > 64 and rng.random() < 0.3) else 0
        act, use_res = int(rng.choice([0, 1, 2])), bool(rng.random() < 0.5)
        g = Geom.conv(h, w, ksize=ks, batch=b)
        descs.append(f"{b}x{h}x{w} {cin - c1}+{c1}->{cout} ks{ks} act {act} res {use_res}")
        x = rnd(b, cin, h, w)
        wt, bias = rnd(cout, cin, ks, ks, scale=(cin * ks * ks) ** -0.5), rnd(cout, scale=0.1)
        pw = gpack(pack_conv(wt, bias))
        rows = x.permute(0, 2, 3, 1).reshape(-1, cin)
        s0, s1 = guarded(rows[:, :cin - c1]), (guarded(rows[:, cin - c1:]) if c1 else None)
        ldo = (cout + 7) // 8 * 8
        res = rnd(g.m, ldo) if use_res else None
        kw = dict(ldo=ldo, act=act, c0=cin - c1, c1=c1)
        if res is not None:
            kw.update(residual=guarded(res), ldr=ldo)
        members.append(((s0, s1, g, pw, guarded(shape=(g.m, ldo))), kw))
        ref = {0: lambda v: v, 1: F.relu, 2: F.silu}[act](F.conv2d(x.float(), wt.float(), bias.float(), padding=ks // 2))
        ref = ref.permute(0, 2, 3, 1).reshape(g.m, cout)
        refs.append((ref + res[:, :cout].float() if res is not None else ref, cout))
    desc = f"group of {n}: tile {tile} pipeline {pipeline} split {split} inkernel {inkernel} | " + " | ".join(descs)
    print(desc, flush=True)
    ops.allocator = lambda nbytes: guarded(shape=(int(nbytes),), dtype=torch.uint8)
    try:
        ops.conv_group(members, form=(tile, split, inkernel, pipeline))
        ops.synchronize()
        got = [a[4].clone() for a, kw in members]
        ops.inkernel_splitk = inkernel
        for (a, kw), g_, (ref, cout) in zip(members, got, refs):
            close(g_[:, :cout], ref, desc)
            a[4].zero_()
            ops.conv(*a, tile=tile, split_k=split, pipeline=pipeline, **kw)
            ops.synchronize()
            assert torch.equal(a[4], g_), "a member launched alone differs from the group: " + desc
    except RuntimeError as e:
        if "failed" in str(e) and "conv_gemm" in str(e):
            return "refused"
        raise
    finally:
        ops.allocator = None
        ops._ws.clear()
        ops.inkernel_splitk = True
    return "ok"


def one_qkv():
    b, hw, c = int(rng.choice([1, 2, 3])), int(rng.integers(1, 300)), int(rng.choice([64, 128, 320]))
    tile = int(rng.choice([0, 1, 2, 3]))
    desc = f"qkv b{b} hw{hw} c{c} tile {tile}"
    print(desc, flush=True)
    m = b * hw
    x = rnd(m, c)
    wt, bias = rnd(3 * c, c, scale=c ** -0.5), rnd(3 * c, scale=0.1)
    pw = gpack(pack_linear(wt, bias))
    t_img = (hw + 63) // 64 * 64
    qk, vt = guarded(shape=(m, 2 * c)), guarded(shape=(c, b * t_img))
    try:
        ops.conv(guarded(x), None, Geom.linear(hw, batch=b), pw, qk, ldo=2 * c, out_t=vt, ldt=b * t_img, t_col0=2 * c, t_img=t_img, tile=tile)
        ops.synchronize()
    except RuntimeError:
        return "refused"
    ref = F.linear(x.float(), wt.float(), bias.float())
    close(qk, ref[:, :2 * c], desc)
    for i in range(b):
        close(vt[:, i * t_img:i * t_img + hw], ref[i * hw:(i + 1) * hw, 2 * c:].t(), desc + f" V^T image {i}")
    return "ok"


def one_xattn():
    """LayerNorm row statistics -> absorbed cross-attention (tile softmax behind split-K: round 4's out-of-bounds read) -> output GEMM"""
    m, c, heads = int(rng.choice([1, 2, 7, 64, 65, 200, 300])), int(rng.choice([640, 1280])), 8
    tile, split = int(rng.choice([0, 3])), int(rng.choice([1, 2, 3, 4]))
    desc = f"xattn m{m} c{c} tile {tile} split {split}"
    print(desc, flush=True)
    tl, d = 77, c // heads
    x = (rnd(m, c).float() * 2 + 0.3).half()
    text_k, text_v = rnd(tl, c, scale=1.5), rnd(tl, c)
    wq, wo, bo = rnd(c, c, scale=c ** -0.5), rnd(c, c, scale=c ** -0.5), rnd(c, scale=0.1)
    gamma, beta = (1 + 0.1 * rnd(c).float()).half(), rnd(c, scale=0.1)
    p0 = gpack(pack_linear(torch.eye(c).half(), None))
    h, rs = guarded(shape=(m, c)), guarded(shape=(m, c // 64, 2), dtype=torch.float32)
    ops.conv(guarded(x), None, Geom.linear(m), p0, h, rowstat_out=rs, tile=2)
    xa1, xa2 = pack_cross_attention(text_k.float(), text_v.float(), wq, wo, bo, gamma, beta, heads)
    xa1, xa2 = gpack(xa1), gpack(xa2)
    pr, out = guarded(shape=(m, heads * 128)), guarded(shape=(m, c))
    ws = guarded(shape=(split * m, heads * 128), dtype=torch.float32) if split > 1 else None
    ops.conv(h, None, Geom.linear(m), xa1, pr, ln_part=rs, act=L.ACT_SOFTMAX, softmax_cols=tl, tile=tile, split_k=split, workspace=ws)
    ops.conv(pr, None, Geom.linear(m), xa2, out, residual=h)
    ops.synchronize()
    xf = x.float()
    ln = F.layer_norm(xf, (c,), gamma.float(), beta.float(), 1e-5)
    q = F.linear(ln, wq.float()).view(m, heads, d).transpose(0, 1)
    kk, vv = text_k.float().view(tl, heads, d).transpose(0, 1), text_v.float().view(tl, heads, d).transpose(0, 1)
    att = torch.softmax(q @ kk.transpose(-1, -2) * d ** -0.5, dim=-1)
    ref = F.linear((att @ vv).transpose(0, 1).reshape(m, c), wo.float(), bo.float()) + xf
    close(out, ref, desc, rel=5e-3)
    return "ok"


def one_groupnorm():
    b, hw = int(rng.choice([1, 2, 5])), int(rng.choice([1, 4, 9, 64, 100, 256, 1024, 1369]))
    c0, c1 = int(rng.choice([64, 320, 640, 1280])), int(rng.choice([0, 0, 64, 320]))
    silu = bool(rng.random() < 0.5)
    desc = f"groupnorm b{b} hw{hw} c{c0}+{c1} silu {silu}"
    print(desc, flush=True)
    c = c0 + c1
    x0, x1 = rnd(b * hw, c0), (rnd(b * hw, c1) if c1 else None)
    gamma, beta = (1 + 0.1 * rnd(c).float()).half(), rnd(c, scale=0.1)
    out = guarded(shape=(b * hw, c))
    ops.groupnorm(guarded(x0), None if x1 is None else guarded(x1), c0, c1, hw, 32, 1e-5, guarded(gamma), guarded(beta), silu, out, batch=b)
    ops.synchronize()
    xin = torch.cat([x0] + ([x1] if x1 is not None else []), dim=1).float().view(b, hw, c).transpose(1, 2)
    ref = F.group_norm(xin, 32, gamma.float(), beta.float(), 1e-5)
    close(out, (F.silu(ref) if silu else ref).transpose(1, 2).reshape(b * hw, c), desc)
    return "ok"


def one_attention():
    b, heads, d = int(rng.choice([1, 2])), int(rng.choice([1, 5, 8])), int(rng.choice([40, 64, 80, 160]))
    sq, sk = int(rng.integers(1, 200)), int(rng.integers(1, 300))
    causal = bool(sq == sk or rng.random() < 0.1) and sq <= sk and b == 1 and rng.random() < 0.3
    desc = f"attention b{b} sq{sq} sk{sk} heads{heads} d{d} causal {causal}"
    print(desc, flush=True)
    c = heads * d
    q, k, v = rnd(b * sq, c), rnd(b * sk, c), rnd(b, sk, c)
    t_img = (sk + 63) // 64 * 64
    vt = torch.zeros(c, b * t_img, dtype=torch.float16)
    for i in range(b):
        vt[:, i * t_img:i * t_img + sk] = v[i].t()
    out = guarded(shape=(b * sq, c))
    ops.attention(guarded(q), c, guarded(k), c, guarded(vt), b * t_img, out, c, sq, sk, heads, d, d ** -0.5, causal=causal, batch=b, k_brows=sk, vt_bcols=t_img)
    ops.synchronize()
    for i in range(b):
        qi = q[i * sq:(i + 1) * sq].float().view(sq, heads, d).transpose(0, 1)
        ki = k[i * sk:(i + 1) * sk].float().view(sk, heads, d).transpose(0, 1)
        vi = v[i].float().view(sk, heads, d).transpose(0, 1)
        s = qi @ ki.transpose(-1, -2) * d ** -0.5
        if causal:
            s = s + torch.full((sq, sk), float("-inf")).triu(1)
        close(out[i * sq:(i + 1) * sq], (torch.softmax(s, dim=-1) @ vi).transpose(0, 1).reshape(sq, c), desc + f" image {i}", rel=6e-3)
    return "ok"


def one_layernorm():
    rows, c = int(rng.integers(1, 500)), int(rng.choice([64, 128, 320, 768, 1280]))
    print(f"layernorm {rows}x{c}", flush=True)
    x = (rnd(rows, c).float() * 2 + 0.5).half()
    gamma, beta = (1 + 0.1 * rnd(c).float()).half(), rnd(c, scale=0.1)
    out = guarded(shape=(rows, c))
    ops.layernorm(guarded(x), rows, c, guarded(gamma), guarded(beta), 1e-5, out)
    ops.synchronize()
    close(out, F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), 1e-5), f"layernorm {rows}x{c}")
    return "ok"


_TAIL = {}


def one_tail():
    """the fused transformer tails (csrc/fused_tail.hip, C = 320) at ragged token counts"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ops_gpu import _tail_weights

    m = int(rng.choice([1, 7, 64, 65, 200, 1000, 3600, 4097, 8000]))
    print(f"tails m{m}", flush=True)
    c = 320
    if not _TAIL:
        w, packs = _tail_weights(c)
        _TAIL["w"], _TAIL["pk"] = w, {k: gpack(v) for k, v in packs.items()}
        _TAIL["keep"] = list(live)
    w, pk = _TAIL["w"], _TAIL["pk"]
    att, h, x, att2 = rnd(m, c), (rnd(m, c).float() * 2 + 0.5).half(), rnd(m, c), rnd(m, c)
    h1, q, out = guarded(shape=(m, c)), guarded(shape=(m, c)), guarded(shape=(m, c))
    ops.tail_a(guarded(att), guarded(h), m, pk["out1"], pk["q2"], h1, q)
    ops.tail_b(guarded(att2), h1, guarded(x), m, pk["out2"], pk["ff1"], pk["ff2"], pk["proj"], out)
    ops.synchronize()
    h1_ref = F.linear(att.float(), w["wo1"].float(), w["bo1"].float()) + h.float()
    close(h1, h1_ref, f"tail_a h1 m{m}")
    h2 = F.linear(att2.float(), w["wo2"].float(), w["bo2"].float()) + h1.float().cpu()
    ln = F.layer_norm(h2.half().float(), (c,), w["g3"].float(), w["be3"].float(), 1e-5)
    hid, gate = F.linear(ln, w["wf1"].float(), w["bf1"].float()).chunk(2, dim=-1)
    h3 = F.linear(hid * F.gelu(gate), w["wf2"].float(), w["bf2"].float()) + h2
    close(out, F.linear(h3, w["wp"].float().reshape(c, c), w["bp"].float()) + x.float(), f"tail_b m{m}")
    return "ok"


def one_geglu():
    from videosd_amd.packing import pack_geglu

    m, c, tile = int(rng.choice([1, 63, 200, 500])), int(rng.choice([320, 640])), int(rng.choice([0, 3]))
    print(f"geglu m{m} c{c} tile {tile}", flush=True)
    x, wt, b = rnd(m, c), rnd(8 * c, c, scale=c ** -0.5), rnd(8 * c, scale=0.1)
    out = guarded(shape=(m, 4 * c))
    ops.conv(guarded(x), None, Geom.linear(m), gpack(pack_geglu(wt, b)), out, tile=tile)
    ops.synchronize()
    hid, gate = F.linear(x.float(), wt.float(), b.float()).chunk(2, dim=-1)
    close(out, hid * F.gelu(gate), f"geglu m{m} c{c}")
    return "ok"


def one_pixels():
    """u8 frame -> fp16 rows, Sobel control map, fp16 rows -> u8 frame at ragged sizes"""
    h, w = int(rng.integers(1, 70)) * 8, int(rng.integers(1, 70)) * 8
    print(f"pixels {h}x{w}", flush=True)
    f = torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    frame = guarded(f)
    rows = guarded(shape=(h * w, 8))
    ops.preprocess_rgb(frame, h, w, rows)
    edge, ctrl = guarded(shape=(h * w,), dtype=torch.uint8), guarded(shape=(h * w, 8))
    ops.sobel_control(frame, h, w, 0.11, 0.8, edge, ctrl)
    img = guarded(shape=(h, w, 3), dtype=torch.uint8)
    ops.postprocess_rgb(rows, 8, h * w, img)
    ops.synchronize()
    want = torch.round(((f.float() / 255 * 2 - 1).half().float() / 2 + 0.5).clamp(0, 1) * 255)
    assert float((img.cpu().float() - want).abs().max()) <= 1.0, "pixel round trip"
    return "ok"


if "selftest" in sys.argv:
    t = guarded(shape=(1024,), dtype=torch.float32, fill=1.0)
    ptr = live[-1]
    print("selftest: reading 64 KB past a guarded buffer (both directions are covered by where it sits): must fault", flush=True)
    big = torch.as_tensor(_Iface(ptr - 65536, (2 * 65536 // 4 + 1024,), torch.float32), device="cuda")
    print(float(big.sum()), "NO FAULT: the guard pages do not work here", flush=True)
    sys.exit(3)

kinds = [one_conv, one_conv, one_conv, one_group, one_group, one_qkv, one_xattn, one_groupnorm, one_attention, one_layernorm, one_pixels, one_tail, one_geglu]
count = {}
t_end = time.time() + seconds
while time.time() < t_end:
    f = kinds[int(rng.integers(len(kinds)))]
    r = f()
    release()
    count[(f.__name__, r)] = count.get((f.__name__, r), 0) + 1
print("guard page fuzz passed:", {f"{k[0]}:{k[1]}": v for k, v in sorted(count.items())})
