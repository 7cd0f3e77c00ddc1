"""Out-of-bounds WRITES: random ragged shapes through the C-ABI ops with every buffer a kernel may write (output, split-K
workspace, transposed V^T output, row statistics) cut out of a larger allocation whose surroundings hold a canary pattern; after
every call the canaries must be intact and the result must match the fp32 reference.  (Out-of-bounds READS do not show here:
they fault or not by what the allocator placed behind the buffer -- scripts/big_frames.py / option_fuzz.py look for those.)
usage (GPU box): python scripts/guard_fuzz.py [seconds=90] [seed=0]"""
import os, sys, time
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videosd_amd import lib as L
from videosd_amd.ops import Geom, HipOps
from videosd_amd.packing import pack_conv, pack_linear

GUARD = 8192  # bytes on either side
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ops = HipOps(0)


class Guarded:
    """a [rows, cols] tensor of `dtype` inside a byte buffer with canaries before and after"""

    def __init__(self, rows, cols, dtype=torch.float16, fill=0.0):
        esz = torch.empty(0, dtype=dtype).element_size()
        self.nbytes = rows * cols * esz
        pad = (-self.nbytes) % 256
        self.raw = torch.full((GUARD + self.nbytes + pad + GUARD,), 0xA5, dtype=torch.uint8, device="cuda")
        self.t = self.raw[GUARD:GUARD + self.nbytes].view(dtype).view(rows, cols)
        self.t.fill_(fill)
        self.tail0 = GUARD + self.nbytes

    def intact(self):
        return bool((self.raw[:GUARD] == 0xA5).all()) and bool((self.raw[self.tail0:] == 0xA5).all())


def rnd(*shape, scale=1.0):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * scale).half()


def close(got, ref, what, rel=4e-3):
    got, ref = got.float().cpu(), ref.float()
    assert torch.isfinite(got).all(), what
    l2 = float((got - ref).norm() / (ref.norm() + 1e-9))
    assert l2 <= rel, f"{what}: rel-L2 {l2:.3g}"


def one_conv():
    ks = int(rng.choice([1, 3]))
    stride = int(rng.choice([1, 1, 2])) if ks == 3 else 1
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    b = int(rng.choice([1, 1, 2, 3]))
    cin = int(rng.choice([64, 128, 192, 320, 640]))
    cout = int(rng.choice([8, 24, 64, 72, 136, 200, 320]))
    tile = int(rng.choice([0, 1, 2, 3, 4, 5]))
    pipeline = int(rng.choice([0, 3, 4, 5, 6, 7, 8, 10]))
    split = int(rng.choice([1, 1, 2, 3, 5]))
    inkernel = bool(rng.random() < 0.5)
    act = int(rng.choice([0, 1, 2, 4, 6]))
    use_res = bool(rng.random() < 0.5)
    if pipeline == 10:  # the persistent 64 -> 64 channel form (csrc/conv_c64.hip): the layer it exists for, on ragged images
        ks, stride, cin, cout, tile, split, act = 3, 1, 64, 64, 5, 1, int(rng.choice([0, 1, 2]))
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    x = rnd(b, cin, h, w)
    wt = rnd(cout, cin, ks, ks, scale=(cin * ks * ks) ** -0.5)
    bias = rnd(cout, scale=0.1)
    g = Geom.conv(h, w, ksize=ks, stride=stride, batch=b)
    kt = (cin * ks * ks + 63) // 64
    split = min(split, kt)
    pw = ops.to_device_pack(pack_conv(wt, bias))
    src = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().cuda()
    ldo = (cout + 7) // 8 * 8
    out = Guarded(g.m, ldo)
    ws = Guarded(max(1, split) * g.m, cout, dtype=torch.float32)
    res = rnd(g.m, ldo) if use_res else None
    desc = f"conv {b}x{h}x{w} {cin}->{cout} ks{ks} s{stride} tile {tile} pipeline {pipeline} split {split} inkernel {inkernel} act {act} res {use_res}"
    ops.inkernel_splitk = inkernel
    try:
        ops.conv(src, None, g, pw, out.t, ldo=ldo, act=act, residual=None if res is None else res.cuda(), ldr=ldo,
                 tile=tile, split_k=split, pipeline=pipeline, workspace=ws.t if split > 1 else None)
        ops.synchronize()
    except RuntimeError:
        return "refused"
    finally:
        ops.inkernel_splitk = True
    assert out.intact(), "output canary overwritten: " + desc
    assert ws.intact(), "workspace canary overwritten: " + desc
    ref = F.conv2d(x.float(), wt.float(), bias.float(), stride=stride, padding=ks // 2)
    ref = {0: lambda v: v, 1: F.relu, 2: F.silu, 4: lambda v: v * torch.sigmoid(1.702 * v), 6: F.gelu}[act](ref)
    ref = ref.permute(0, 2, 3, 1).reshape(g.m, cout)
    if res is not None:
        ref = ref + res[:, :cout].float()
    close(out.t[:, :cout], ref, desc)
    return "ok"


def one_qkv():
    """the transposed-output epilogue (V^T slabs per image) + row statistics"""
    b = int(rng.choice([1, 2, 3]))
    hw = int(rng.integers(1, 300))
    c = int(rng.choice([64, 128, 320]))
    m = b * hw
    x = rnd(m, c)
    wt, bias = rnd(3 * c, c, scale=c ** -0.5), rnd(3 * c, scale=0.1)
    pw = ops.to_device_pack(pack_linear(wt, bias))
    t_img = (hw + 63) // 64 * 64
    qk = Guarded(m, 2 * c)
    vt = Guarded(c, b * t_img)
    tile = int(rng.choice([0, 1, 2, 3]))
    desc = f"qkv b{b} hw{hw} c{c} tile {tile}"
    try:
        ops.conv(x.cuda(), None, Geom.linear(hw, batch=b), pw, qk.t, ldo=2 * c, out_t=vt.t, ldt=b * t_img, t_col0=2 * c, t_img=t_img, tile=tile)
        ops.synchronize()
    except RuntimeError:
        return "refused"
    assert qk.intact() and vt.intact(), "canary overwritten: " + desc
    ref = F.linear(x.float(), wt.float(), bias.float())
    close(qk.t, ref[:, :2 * c], desc)
    for i in range(b):
        close(vt.t[:, i * t_img:i * t_img + hw], ref[i * hw:(i + 1) * hw, 2 * c:].t(), desc + f" V^T image {i}")
    return "ok"


def one_groupnorm():
    b = int(rng.choice([1, 2, 5]))
    hw = int(rng.choice([1, 4, 9, 64, 100, 256, 1024, 1369]))
    c0 = int(rng.choice([64, 320, 640, 1280]))
    c1 = int(rng.choice([0, 0, 64, 320]))
    c = c0 + c1
    x0, x1 = rnd(b * hw, c0), (rnd(b * hw, c1) if c1 else None)
    gamma, beta = (1 + 0.1 * rnd(c).float()).half(), rnd(c, scale=0.1)
    out = Guarded(b * hw, c)
    silu = bool(rng.random() < 0.5)
    desc = f"groupnorm b{b} hw{hw} c{c0}+{c1} silu {silu}"
    ops.groupnorm(x0.cuda(), None if x1 is None else x1.cuda(), c0, c1, hw, 32, 1e-5, gamma.cuda(), beta.cuda(), silu, out.t, batch=b)
    ops.synchronize()
    assert out.intact(), "canary overwritten: " + desc
    xin = torch.cat([x0] + ([x1] if x1 is not None else []), dim=1).float().view(b, hw, c).transpose(1, 2)
    ref = F.group_norm(xin, 32, gamma.float(), beta.float(), 1e-5)
    ref = (F.silu(ref) if silu else ref).transpose(1, 2).reshape(b * hw, c)
    close(out.t, ref, desc)
    return "ok"


def one_attention():
    b = int(rng.choice([1, 2]))
    heads, d = int(rng.choice([1, 5, 8])), int(rng.choice([40, 64, 80, 160]))
    sq, sk = int(rng.integers(1, 200)), int(rng.integers(1, 300))
    c = heads * d
    q, k = rnd(b * sq, c), rnd(b * sk, c)
    t_img = (sk + 63) // 64 * 64
    v = rnd(b, sk, c)
    vt = torch.zeros(c, b * t_img, dtype=torch.float16)
    for i in range(b):
        vt[:, i * t_img:i * t_img + sk] = v[i].t()
    out = Guarded(b * sq, c)
    desc = f"attention b{b} sq{sq} sk{sk} heads{heads} d{d}"
    ops.attention(q.cuda(), c, k.cuda(), c, vt.cuda(), b * t_img, out.t, c, sq, sk, heads, d, d ** -0.5, batch=b, k_brows=sk, vt_bcols=t_img)
    ops.synchronize()
    assert out.intact(), "canary overwritten: " + desc
    for i in range(b):
        qi = q[i * sq:(i + 1) * sq].float().view(sq, heads, d).transpose(0, 1)
        ki = k[i * sk:(i + 1) * sk].float().view(sk, heads, d).transpose(0, 1)
        vi = v[i].float().view(sk, heads, d).transpose(0, 1)
        ref = (torch.softmax(qi @ ki.transpose(-1, -2) * d ** -0.5, dim=-1) @ vi).transpose(0, 1).reshape(sq, c)
        close(out.t[i * sq:(i + 1) * sq], ref, desc + f" image {i}", rel=6e-3)
    return "ok"


def one_layernorm():
    rows, c = int(rng.integers(1, 500)), int(rng.choice([64, 128, 320, 768, 1280]))
    x = (rnd(rows, c).float() * 2 + 0.5).half()
    gamma, beta = (1 + 0.1 * rnd(c).float()).half(), rnd(c, scale=0.1)
    out = Guarded(rows, c)
    ops.layernorm(x.cuda(), rows, c, gamma.cuda(), beta.cuda(), 1e-5, out.t)
    ops.synchronize()
    assert out.intact(), f"canary overwritten: layernorm {rows}x{c}"
    close(out.t, F.layer_norm(x.float(), (c,), gamma.float(), beta.float(), 1e-5), f"layernorm {rows}x{c}")
    return "ok"


kinds = [one_conv, one_conv, one_conv, one_qkv, one_groupnorm, one_attention, one_layernorm]
count = {}
t_end = time.time() + seconds
while time.time() < t_end:
    f = kinds[int(rng.integers(len(kinds)))]
    r = f()
    count[(f.__name__, r)] = count.get((f.__name__, r), 0) + 1
print("guard fuzz passed:", {f"{k[0]}:{k[1]}": v for k, v in sorted(count.items())})
