"""TEST-ONLY stand-in for videosd_amd.ops.HipOps: the same op interface evaluated with torch on the CPU
(fp32 math, fp16 storage, exactly the buffer/stride/epilogue semantics of include/vsd.h).

It exists so that the engine's HOST logic (weight packing, buffer wiring, skip/concat bookkeeping,
schedule constants) can be checked against the oracle in the GPU-less container.  The product never
imports this module and has no CPU fallback: videosd_amd.ops.HipOps raises without a GPU."""
import torch
import torch.nn.functional as F

from videosd_amd import lib as L


class FakeOps:
    name = "fake-cpu"

    def __init__(self):
        self.device = torch.device("cpu")
        self.tile_override = {}

    def empty(self, *shape, dtype=torch.float16):
        return torch.zeros(*shape, dtype=dtype)

    def zeros(self, *shape, dtype=torch.float16):
        return torch.zeros(*shape, dtype=dtype)

    def to_device(self, t):
        return t.clone()

    def to_device_pack(self, p):
        return p

    def synchronize(self):
        pass

    def zero_(self, t):
        t.zero_()

    def copy_(self, dst, src):
        dst.copy_(src)

    def clone(self, lane=None):
        return FakeOps()

    def use_stream(self, idx):
        pass

    def fork(self):
        pass

    def join(self):
        pass

    def signal(self, name):
        pass

    def wait(self, name):
        pass

    def upload(self, dst, src):
        dst.copy_(src.view(dst.shape))

    def download(self, src):
        return src.clone()

    def download_into(self, dst_cpu, src):
        dst_cpu.view(src.shape).copy_(src)

    @staticmethod
    def _nhwc(t, h, w, c, b=1):
        """[b*h*w][>=c] buffer (possibly a strided view) -> NCHW fp32"""
        return t[:, :c].float().reshape(b, h, w, c).permute(0, 3, 1, 2)

    def conv_group(self, calls, form=None, split=None, default=None):
        for a, kw in calls:
            self.conv(*a, **kw)

    def pair(self, a, b):
        for fn, aa, kk in (a, b):
            fn(*aa, **kk)

    def conv(self, src0, src1, g, w, out, *, ldo=None, c0=None, c1=0, rowvec=None, residual=None, residual2=None,
             ldr=None, out_scale=1.0, act=L.ACT_NONE, out2=None, add2=None, out_t=None, ldt=0, t_col0=0, tile=None,
             split_k=None, workspace=None, pipeline=None, rowstat_out=None, ln_part=None, ln_eps=1e-5,
             chanstat_out=None, t_img=0, out_scale_dev=None, softmax_cols=0):
        if out_scale_dev is not None:
            out_scale = float(out_scale_dev.reshape(-1)[0])
        c0 = c0 if c0 is not None else (w.cin - c1)
        B = g.batch
        x = self._nhwc(src0, g.hs, g.ws, c0, B)
        if src1 is not None and c1:
            x = torch.cat([x, self._nhwc(src1, g.hs, g.ws, c1, B)], dim=1)
        assert x.shape[1] == w.cin, (x.shape, w.cin)
        if (g.hi, g.wi) != (g.hs, g.ws):
            x = F.interpolate(x, size=(g.hi, g.wi), mode="nearest")
        wt = w.weight[:, :w.k].float()
        bias = None if w.bias is None else w.bias.float()
        ln_s, ln_t = w.ln_s, w.ln_t

        def untile(v):  # GEGLU tile packing -> [hidden rows | gate rows]
            f = w.n // 2
            vv = v.reshape(f // 64, 2, 64, *v.shape[1:])
            return torch.cat([vv[:, 0].reshape(f, *v.shape[1:]), vv[:, 1].reshape(f, *v.shape[1:])], dim=0)

        if w.geglu:
            wt = untile(wt)
            bias = None if bias is None else untile(bias)
            if ln_s is not None:
                ln_s, ln_t = untile(ln_s), untile(ln_t)
        wt = wt.reshape(w.n, w.ksize, w.ksize, w.cin).permute(0, 3, 1, 2)
        y = F.conv2d(x, wt, bias, stride=g.stride, padding=g.pad)
        assert y.shape[2:] == (g.ho, g.wo), (y.shape, g)
        y = y.permute(0, 2, 3, 1).reshape(g.m, w.n)
        if ln_part is not None:
            tot = ln_part.float().sum(dim=1)  # [M, 2]
            mean = tot[:, 0] / w.k
            rstd = torch.rsqrt((tot[:, 1] / w.k - mean * mean).clamp_min(0) + ln_eps)
            y = rstd[:, None] * (y - mean[:, None] * ln_s.float()[None, :]) + ln_t.float()[None, :]
        if rowvec is not None:
            y = y + rowvec.float()[None, :]
        post = bool(act & L.ACT_POST)
        a = act & 0xFF
        if w.geglu:
            hid, gate = y.chunk(2, dim=-1)
            y = hid * F.gelu(gate)
            a = L.ACT_NONE

        if a == L.ACT_SOFTMAX:  # row softmax inside every 128-column group over its first softmax_cols columns
            yy = y.reshape(g.m, w.n // 128, 128)
            pr = torch.zeros_like(yy)
            pr[:, :, :softmax_cols] = torch.softmax(yy[:, :, :softmax_cols], dim=-1)
            y = pr.reshape(g.m, w.n)
            a = L.ACT_NONE

        def fa(v):
            if a == L.ACT_RELU:
                return F.relu(v)
            if a == L.ACT_SILU:
                return F.silu(v)
            if a == L.ACT_QUICKGELU:
                return v * torch.sigmoid(1.702 * v)
            if a == L.ACT_GELU:
                return F.gelu(v)
            return v

        if not post:
            y = fa(y)
        y = y * out_scale
        nout = y.shape[1]
        if out_t is not None:
            yt = y[:, t_col0:]
            if B == 1:
                out_t[: yt.shape[1], : g.m] = yt.t().half()
            else:
                hw = g.m // B
                ti = t_img or hw
                assert ti >= hw and B * ti <= out_t.shape[1]
                for b in range(B):
                    out_t[: yt.shape[1], b * ti: b * ti + hw] = yt[b * hw:(b + 1) * hw].t().half()
            y = y[:, :t_col0]
            nout = t_col0
        if residual is not None:
            y = y + residual[:, :nout].float()
        if residual2 is not None:
            y = y + residual2[:, :nout].float()
        if post:
            y = fa(y)
        out[:, :nout] = y.half()
        if rowstat_out is not None:
            o = out[:, :nout].float().reshape(g.m, nout // 64, 64)
            rowstat_out[:, :, 0] = o.sum(dim=2)
            rowstat_out[:, :, 1] = (o * o).sum(dim=2)
        if chanstat_out is not None:
            o = out[:, :nout].float()
            chanstat_out[:, 0] = o.sum(dim=0)
            chanstat_out[:, 1] = (o * o).sum(dim=0)
        if out2 is not None:
            out2[:, :nout] = (y + add2[:, :nout].float()).half()

    def groupnorm(self, src0, src1, c0, c1, hw, groups, eps, gamma, beta, silu, out, batch=1):
        if batch > 1:
            for b in range(batch):
                sl = slice(b * hw, (b + 1) * hw)
                self.groupnorm(src0[sl], None if src1 is None else src1[sl], c0, c1, hw, groups, eps, gamma, beta, silu, out[sl])
            return
        x = src0[:, :c0].float()
        if src1 is not None and c1:
            x = torch.cat([x, src1[:, :c1].float()], dim=1)
        y = F.group_norm(x.t()[None], groups, gamma.float(), beta.float(), eps)[0].t()
        if silu:
            y = F.silu(y)
        out[:, : c0 + c1] = y.half()

    def layernorm(self, x, rows, c, gamma, beta, eps, out):
        out[:, :c] = F.layer_norm(x[:, :c].float(), (c,), gamma.float(), beta.float(), eps).half()

    def attention(self, q, ldq, k, ldk, vt, ldvt, out, ldo, sq, sk, heads, d, scale, causal=False, batch=1, k_brows=0,
                  vt_bcols=0):
        if batch > 1:
            for b in range(batch):
                self.attention(q[b * sq:], ldq, k[b * k_brows:], ldk, vt[:, b * vt_bcols:], ldvt, out[b * sq:], ldo, sq, sk,
                               heads, d, scale, causal, batch=-1)
            return
        c = heads * d
        qh = q[:sq, :c].float().reshape(sq, heads, d).transpose(0, 1)
        kh = k[:sk, :c].float().reshape(sk, heads, d).transpose(0, 1)
        vh = vt[:c, :sk].float().t().reshape(sk, heads, d).transpose(0, 1)
        pad_end = -(-sk // 64) * 64
        assert batch == -1 or (vt[:c, sk:pad_end] == 0).all(), "V^T key padding must be zero"
        assert torch.isfinite(vt[:c, sk:pad_end].float()).all()
        s = qh @ kh.transpose(-1, -2) * scale
        if causal:
            s = s + torch.full((sq, sk), float("-inf")).triu(1)
        o = (torch.softmax(s, dim=-1) @ vh).transpose(0, 1).reshape(sq, c)
        out[:sq, :c] = o.half()

    def preprocess_rgb(self, rgb_u8, h, w, out):
        x = rgb_u8.reshape(h * w, 3).float() / 255.0  # h may be batch * H: purely per pixel
        y = (2.0 * x - 1.0).half()
        out.zero_()
        out[:, :3] = ((y.float() + 1.0).half().float() * 0.5).half()

    def sobel_control(self, rgb_u8, h, w, low, high, edge_u8, control_out):
        p = rgb_u8.reshape(h, w, 3).to(torch.int64)
        lum = (p[..., 0] * 19595 + p[..., 1] * 38470 + p[..., 2] * 7471 + 0x8000) >> 16
        x = (lum.float() / 255.0)[None, None]
        kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]).view(1, 1, 3, 3)
        ky = torch.tensor([[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]]).view(1, 1, 3, 3)
        e = torch.sqrt(F.conv2d(x, kx, padding=1) ** 2 + F.conv2d(x, ky, padding=1) ** 2)
        e = e / e.max()
        e[e >= high] = 1.0
        e[e <= low] = 0.0
        u = torch.nan_to_num(e, nan=0.0).mul(255).byte().reshape(-1)
        edge_u8.copy_(u)
        control_out.zero_()
        v = (u.float() / 255.0).half()
        control_out[:, 0] = v
        control_out[:, 1] = v
        control_out[:, 2] = v

    def add_noise(self, x0, noise_f32, sqrt_a, sqrt_b, hw, out):
        out.zero_()
        out[:, :4] = (sqrt_a * x0[:, :4].float() + sqrt_b * noise_f32.reshape(4, hw).t()).half()

    def adain(self, x, stats, stats_ref, rows, c, out, eps=1e-6):
        mean, mean_r = stats[:, 0] / rows, stats_ref[:, 0] / rows
        var = (stats[:, 1] / rows - mean * mean).clamp_min(0)
        var_r = (stats_ref[:, 1] / rows - mean_r * mean_r).clamp_min(0)
        sd, sd_r = var.clamp_min(eps).sqrt(), var_r.clamp_min(eps).sqrt()
        out[:, :c] = (((x[:, :c].float() - mean) / sd) * sd_r + mean_r).half()

    TAIL_C = 320

    @staticmethod
    def _ln_rows(x16, eps):
        xf = x16.float()
        mean = xf.mean(dim=1, keepdim=True)
        var = (xf * xf).mean(dim=1, keepdim=True) - mean * mean
        return mean, torch.rsqrt(var.clamp_min(0) + eps)

    def tail_a(self, att, h, m, out1, q2, h1_out, q_out, ln_eps=1e-5):
        h1 = att[:m].float() @ out1.weight[:, :out1.k].float().t() + out1.bias.float() + h[:m].float()
        h1_out[:m] = h1.half()
        mean, rstd = self._ln_rows(h1_out[:m], ln_eps)
        acc = h1_out[:m].float() @ q2.weight[:, :q2.k].float().t()
        q_out[:m] = (rstd * (acc - mean * q2.ln_s.float()[None, :]) + q2.ln_t.float()[None, :]).half()

    def tail_b(self, att2, h1, x, m, out2, ff1, ff2, proj, out, ln_eps=1e-5):
        h2 = att2[:m].float() @ out2.weight[:, :out2.k].float().t() + out2.bias.float() + h1[:m].float()
        h2h = h2.half()
        mean, rstd = self._ln_rows(h2h, ln_eps)
        acc = h2h.float() @ ff1.weight[:, :ff1.k].float().t()
        y = rstd * (acc - mean * ff1.ln_s.float()[None, :]) + ff1.ln_t.float()[None, :]
        yy = y.reshape(m, ff1.n // 128, 2, 64)
        hid = (yy[:, :, 0] * F.gelu(yy[:, :, 1])).reshape(m, ff1.n // 2).half()
        h3 = hid.float() @ ff2.weight[:, :ff2.k].float().t() + ff2.bias.float() + h2
        out[:m] = (h3.half().float() @ proj.weight[:, :proj.k].float().t() + proj.bias.float() + x[:m].float()).half()

    def xattn_fold(self, k, vt, tl, wq, wo, gamma, beta, c, heads, scale, xa1_w, xa1_s, xa1_t, xa2_w):
        """vsd_xattn_fold through the host restatement of the same algebra (packing.pack_cross_attention)"""
        from videosd_amd.packing import pack_cross_attention

        x1, x2 = pack_cross_attention(k[:tl, :c].float(), vt[:c, :tl].float().t().contiguous(), wq[:, :c], wo[:, :c],
                                      torch.zeros(c), gamma, beta, heads)
        xa1_w.copy_(x1.weight[:, :c])
        xa1_s.copy_(x1.ln_s)
        xa1_t.copy_(x1.ln_t)
        xa2_w.copy_(x2.weight[:, :heads * 128])

    def embed_tokens(self, ids_i64, tok_emb, pos_emb, out):
        n = out.shape[0]
        out.copy_((tok_emb[ids_i64.long()].float() + pos_emb[:n].float()).half())

    def add_noise_dev(self, x0, noise_f32, coef_dev, hw, batch, out):
        sa, sb = [float(v) for v in coef_dev.reshape(-1)[:2]]
        for b in range(batch):
            self.add_noise(x0[b * hw:(b + 1) * hw], noise_f32, sa, sb, hw, out[b * hw:(b + 1) * hw])

    def lcm_step_dev(self, eps, sample, noise_f32, coef_dev, hw, batch, prev, denoised, dec_in=None):
        coef = [float(v) for v in coef_dev.reshape(-1)[:6]]
        sl = lambda t, b: None if t is None else t[b * hw:(b + 1) * hw]  # noqa: E731
        for b in range(batch):
            self.lcm_step(sl(eps, b), sl(sample, b), noise_f32, coef, hw, sl(prev, b), sl(denoised, b), sl(dec_in, b))

    def lcm_step(self, eps, sample, noise_f32, coef, hw, prev, denoised, dec_in=None):
        sa, sb, cskip, cout, sap, sbp = [float(torch.tensor(c, dtype=torch.float32)) for c in coef]
        xs = sample[:, :4].float()
        px0 = (xs - sb * eps[:, :4].float()) / sa
        d = cout * px0 + cskip * xs
        pv = d if noise_f32 is None else sap * d + sbp * noise_f32.reshape(4, hw).t()
        if prev is not None:
            prev.zero_()
            prev[:, :4] = pv.half()
        if denoised is not None:
            denoised.zero_()
            denoised[:, :4] = d.half()
        if dec_in is not None:
            dec_in.zero_()
            dec_in[:, :4] = (torch.tanh(d.half().float() / 3.0) * 3.0).half()

    def postprocess_rgb(self, img, ld, hw, rgb_u8):
        y = (img[:, :3].float() * 2.0 - 1.0).half()
        z = (y.float() * 0.5 + 0.5).half().float().clamp(0, 1)
        rgb_u8.reshape(hw, 3).copy_((z * 255.0).round().to(torch.uint8))

    def axpy(self, a, b, scale, n, out):
        out.copy_((a.float() + scale * b.float()).half())

    def graph_begin(self):
        raise RuntimeError("FakeOps has no graphs")

    def graph_destroy(self, g):
        pass

    # launch sequences (HipOps.seq_*): the emulator keeps the STRUCTURE the engine builds -- ("graph", stream, calls run while
    # capturing) / ("record", stream, event) / ("wait", stream, event) -- so that Engine._capture is testable without a GPU
    def seq_create(self):
        return {"items": [], "events": 0, "open": None}

    def seq_capture_begin(self, sidx):
        self._calls = 0

    def seq_capture_end(self, seq, sidx):
        seq["items"].append(("graph", sidx))

    def seq_record(self, seq, sidx):
        seq["items"].append(("record", sidx, seq["events"]))
        seq["events"] += 1
        return seq["events"] - 1

    def seq_wait(self, seq, sidx, ev):
        assert 0 <= ev < seq["events"]
        seq["items"].append(("wait", sidx, ev))

    def seq_count(self, seq):
        return sum(i[0] == "graph" for i in seq["items"]), sum(i[0] == "wait" for i in seq["items"])

    def seq_launch(self, seq):
        raise RuntimeError("FakeOps cannot launch a sequence")

    def seq_destroy(self, seq):
        seq["items"] = None
