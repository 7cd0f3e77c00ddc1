"""HipOps: the op interface the engine is written against, bound to libvsd.so through ctypes.

Tensors are torch CUDA(ROCm) tensors used purely as device buffers (`.data_ptr()`); every compute
call goes through the C-ABI of include/vsd.h on `self.stream`.
"""
import ctypes as C
import math
import time
from dataclasses import dataclass
from typing import Optional

import torch

from . import lib as L
from .packing import PackedConv

import os as _os

POISON = bool(_os.environ.get("VSD_POISON"))


@dataclass(frozen=True)
class Geom:
    """Spatial geometry of one implicit-GEMM conv: stored source size, logical (resized) input size, output size."""
    hs: int
    ws: int
    hi: int
    wi: int
    ho: int
    wo: int
    ksize: int = 1
    stride: int = 1
    pad: int = 0
    batch: int = 1  # images stacked along M (frames of independent streams sharing one launch)

    @staticmethod
    def linear(rows: int, batch: int = 1) -> "Geom":
        """`rows` per image; a plain GEMM does not care where one image ends (batch only matters to a transposed
        output, which keeps each image in its own column slab)."""
        return Geom(1, rows, 1, rows, 1, rows, 1, 1, 0, batch)

    @staticmethod
    def conv(h: int, w: int, ksize=3, stride=1, up_to=None, batch=1) -> "Geom":
        hi, wi = (h, w) if up_to is None else up_to
        pad = ksize // 2
        ho = (hi + 2 * pad - ksize) // stride + 1
        wo = (wi + 2 * pad - ksize) // stride + 1
        return Geom(h, w, hi, wi, ho, wo, ksize, stride, pad, batch)

    @property
    def m(self) -> int:
        return self.batch * self.ho * self.wo


def choose_tile(m: int, n: int, kp: int, geglu: bool = False, t_col0: int = 0):
    """Heuristic (tile, split_k): minimise padding waste, prefer big tiles, then split K until the grid
    covers the 256 CUs.  The engine's autotuner can override this per layer."""
    cands = [L.TILE_128x128, L.TILE_64x128] if geglu else [L.TILE_128x128, L.TILE_128x64, L.TILE_64x128, L.TILE_64x64]
    pen = {L.TILE_128x128: 1.0, L.TILE_128x64: 1.12, L.TILE_64x128: 1.12, L.TILE_64x64: 1.3}
    kt = kp // 64
    best = None
    for t in cands:
        bm, bn = L.TILE_DIMS[t]
        if t_col0 % bn:
            continue
        tm, tn = -(-m // bm), -(-n // bn)
        blocks = tm * tn
        waste = (tm * bm * tn * bn) / float(m * n)
        split = 1
        if not geglu and blocks < 256 and kt >= 8:
            split = max(1, min(-(-320 // blocks), kt // 4))
        total = blocks * split
        fill = min(1.0, total / 256.0)
        # rounds of 256-CU waves at ~2 blocks/CU
        cost = waste * pen[t] / max(fill, 0.05) * (1.0 + 0.04 * (split - 1))
        if best is None or cost < best[0]:
            best = (cost, t, split)
    return best[1], best[2]


class LaneBook:
    """What the HipOps objects of one process and device share about launch lanes: which lanes have a live object (weak
    references: a dropped slot gives its lane back), and the helper objects the throughput-mode tuner times candidates with."""

    def __init__(self):
        import weakref

        self._weakref = weakref
        self.users = {}          # lane -> WeakSet of HipOps
        self.tune_helpers = {}   # lane -> HipOps kept for HipOps._time_candidates_on_lanes (made once per process)
        self.warned = False

    def register(self, ops: "HipOps"):
        self.users.setdefault(ops.lane, self._weakref.WeakSet()).add(ops)

    def next_free(self) -> int:
        """the lowest lane no live object runs on; when all four launch streams have one, lanes are shared round-robin (their
        objects then take turns on a stream: said once, since engines meant to run side by side would not)"""
        for lane in range(L.POOL_STREAMS):
            if not len(self.users.get(lane, ())):
                return lane
        if not self.warned:
            self.warned = True
            import warnings

            warnings.warn(f"HipOps.clone(): all {L.POOL_STREAMS} launch lanes have a live object; further auto-numbered clones share a lane's "
                          "streams with an existing one and take turns with it (pass lane= to choose)")
        return min(range(L.POOL_STREAMS), key=lambda l: len(self.users.get(l, ())))


class HipOps:
    name = "hip"

    def __init__(self, device_id: int = 0, stream: Optional[torch.cuda.Stream] = None, make_current: bool = True,
                 tile_override: Optional[dict] = None, lane: int = 0, _lanes: Optional[LaneBook] = None, _register: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("HipOps needs a ROCm GPU; there is no CPU fallback in the product path")
        self.device = torch.device("cuda", device_id)
        torch.cuda.set_device(self.device)
        self.ctx = L.Context(device_id)
        # Launch lanes.  Everything that can be in flight at once on this GPU must sit on a hardware queue AND a command-processor
        # pipe of its own; with plain streams both are accidents of what the process created before (DESIGN.md section 3,
        # "launches in flight").  libvsd keeps four launch streams per process and device (vsd_stream_pool: four queues on four
        # pipes); lane l launches on pool stream l mod 4 and runs its side branch (the ControlNet encoder beside the UNet
        # encoder) on stream (l + 2) mod 4: two lanes with side branches, or four lanes without, never share a pipe.
        # VSD_STREAMS=plain takes torch's pool streams instead (A/B measurements only).
        self.lane = int(lane)
        self._lanes = _lanes if _lanes is not None else LaneBook()  # shared with every clone
        self._plain = _os.environ.get("VSD_STREAMS") == "plain"
        if stream is not None:
            self.streams = [stream, torch.cuda.Stream(device=self.device)]
        elif self._plain:
            self.streams = [torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)]
        else:
            pool = self.pool_streams()
            self.streams = [pool[self.lane % L.POOL_STREAMS], pool[(self.lane + 2) % L.POOL_STREAMS]]
        self.stream = self.streams[0]
        if _register:  # (the tuner's helper objects do not occupy a lane)
            self._lanes.register(self)
        # torch-side plumbing (allocation fills, H2D/D2H copies) must be ordered with the kernels: make the
        # kernel stream this thread's current torch stream.
        if make_current:
            torch.cuda.set_stream(self.stream)
        self.device_id = device_id
        self._sidx = 0
        self._pair_default = None
        self._widx = None  # scratch set of the call being issued when it is not the stream's own (members of a pair / group)
        self._events = {}
        self._ws = {}
        self.tile_override = {} if tile_override is None else tile_override
        # 0: kernel choices timed alone (latency), 1: timed with four lanes busy (throughput).  Set by Engine.prepare for the plan
        # it records / captures; part of the tuning key.
        self.tune_mode = 0
        # timing with four lanes busy takes ~80 s per plan: done offline (scripts/retune_all.py sets this) or on request
        # (VSD_TUNE_LANES=1); a live worker that meets a shape the table has no throughput-mode entry for takes the alone-timed
        # choice (a second or two per plan, as before) instead of stalling a stream for minutes
        self.tune_lanes_online = bool(_os.environ.get("VSD_TUNE_LANES"))
        self.inkernel_splitk = True
        self.one_launch_bias_us = float(_os.environ.get("VSD_ONE_LAUNCH_BIAS_US", "0.5"))  # tune_conv: see there
        self.no_halo = bool(__import__("os").environ.get("VSD_NO_HALO"))  # debugging: run halo-tuned shapes on the generic ring
        self.default_pipeline = 3
        self.no_c64 = bool(_os.environ.get("VSD_NO_C64"))  # (development: keep the persistent 64-channel conv form out of the tuner / the plans)
        self.no_w8 = bool(_os.environ.get("VSD_NO_W8"))  # (development: keep the eight-wave conv forms out of the tuner)
        with torch.cuda.stream(self.stream):
            self._counters = [torch.zeros(L.SPLITK_MAX_TILES, dtype=torch.int32, device=self.device) for _ in range(2)]
        with torch.cuda.stream(self.stream):
            self._chan_counters = [torch.zeros(4096, dtype=torch.int32, device=self.device) for _ in range(2)]
        self.stream.synchronize()

    # ------------------------------------------------------------------ helpers
    def pool_streams(self):
        """the process's four launch streams on this device, as torch streams (include/vsd.h vsd_stream_pool)"""
        h = (C.c_void_p * L.POOL_STREAMS)()
        self.ctx.call("vsd_stream_pool", h)
        return [torch.cuda.ExternalStream(int(h[i]), device=self.device) for i in range(L.POOL_STREAMS)]

    def pool_check(self, chain: int = 100) -> float:
        """time of four frame-like kernel chains on the four launch streams at once / one chain alone: ~1.0 when they run side
        by side, >= 2 when two of them share a command-processor pipe (vsd_stream_pool_check)"""
        r = C.c_float()
        self.ctx.call("vsd_stream_pool_check", int(chain), C.byref(r))
        return float(r.value)

    def _stream(self, idx: int):
        return self.streams[idx]

    @property
    def s(self):
        return C.c_void_p(self._stream(self._sidx).cuda_stream)

    def raw_stream(self, idx: int = 0):
        return C.c_void_p(self._stream(idx).cuda_stream)

    # ---- two-stream fork/join.  Run live (eager mode) these are events between the two streams; in a captured program the
    #      engine turns them into the edges of a launch sequence (Engine._capture), never into branches of one graph.
    def use_stream(self, idx: int):
        self._sidx = idx

    def fork(self):
        e = torch.cuda.Event()
        e.record(self._stream(0))
        self._stream(1).wait_event(e)

    def join(self):
        e = torch.cuda.Event()
        e.record(self._stream(1))
        self._stream(0).wait_event(e)

    # ---- named dependencies between the two streams (edges of the captured graph): `signal` marks a point of the
    #      current stream, `wait` makes the current stream wait for it
    def signal(self, name: str):
        e = torch.cuda.Event()
        e.record(self._stream(self._sidx))
        self._events[name] = e

    def wait(self, name: str):
        self._stream(self._sidx).wait_event(self._events[name])

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def clone(self, lane: Optional[int] = None, _register: bool = True) -> "HipOps":
        """Same GPU, own context / scratch, shared tuning table, the streams of launch lane `lane` (default: the next unused
        one): a further frame in flight.  Objects of ONE lane share its streams -- they are meant to take turns (the engines of a
        lane's several plans / batch sizes); objects of different lanes run side by side."""
        if lane is None:
            lane = self._lanes.next_free()
        return HipOps(self.device_id, make_current=False, tile_override=self.tile_override, lane=int(lane), _lanes=self._lanes,
                      _register=_register)

    # Debugging hook (scripts/guard_page_engine.py): a callable(nbytes) -> uint8 device tensor.  When set, EVERY buffer this object
    # hands out -- activations (the engine's arena then allocates tensor by tensor), packed weights, uploads, workspaces -- comes
    # from it, e.g. from an allocator that ends each buffer at an unmapped page so that an out-of-bounds access faults.
    allocator = None

    def _alloc(self, shape, dtype):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        raw = self.allocator(max(nbytes, 16))
        return raw[:nbytes].view(dtype).view(*shape)

    def empty(self, *shape, dtype=torch.float16):
        if self.allocator is not None:
            with torch.cuda.stream(self.stream):
                t = self._alloc(shape, dtype)
                t.view(torch.uint8).fill_(255 if POISON else 0)
                return t
        with torch.cuda.stream(self.stream):
            t = torch.empty(*shape, dtype=dtype, device=self.device)
            if POISON:  # VSD_POISON=1 (tests): every "uninitialised" byte is 0xFF (fp16 / fp32 NaN), so that a kernel
                t.view(torch.uint8).fill_(255)  # relying on what the allocator happened to hand out fails every time
            return t

    def zeros(self, *shape, dtype=torch.float16):
        with torch.cuda.stream(self.stream):
            if self.allocator is not None:
                t = self._alloc(shape, dtype)
                t.view(torch.uint8).zero_()
                return t
            return torch.zeros(*shape, dtype=dtype, device=self.device)

    def to_device(self, t: torch.Tensor):
        with torch.cuda.stream(self.stream):
            if self.allocator is not None:
                d = self._alloc(tuple(t.shape), t.dtype)
                d.copy_(t.contiguous())
                return d
            return t.to(self.device)

    def zero_(self, t: torch.Tensor):
        with torch.cuda.stream(self.stream):
            t.zero_()

    def copy_(self, dst: torch.Tensor, src: torch.Tensor):
        """device-to-device copy ordered on this ops' kernel stream"""
        with torch.cuda.stream(self.stream):
            dst.copy_(src, non_blocking=True)

    def to_device_pack(self, p: PackedConv) -> PackedConv:
        for f in ("weight", "bias", "ln_s", "ln_t", "weight_frag"):
            v = getattr(p, f)
            if v is not None:
                setattr(p, f, self.to_device(v) if self.allocator is not None else v.to(self.device).contiguous())
        return p

    @property
    def _scratch(self) -> int:
        return self._sidx if self._widx is None else self._widx

    def _counter_set(self, sets, n):
        """split-K / channel-statistics counters of the current scratch set (zeroed once; the kernels leave them at zero)"""
        while len(sets) <= self._scratch:
            with torch.cuda.stream(self.stream):
                sets.append(torch.zeros(n, dtype=torch.int32, device=self.device))
            self.stream.synchronize()
        return sets[self._scratch]

    def workspace(self, key: str, nbytes: int) -> torch.Tensor:
        key = (key, self._scratch)  # kernels on different streams / members of one grid may run concurrently: separate scratch
        cur = self._ws.get(key)
        if cur is None or cur.numel() < nbytes:
            with torch.cuda.stream(self.stream):
                if self.allocator is not None:
                    cur = self.allocator(max(nbytes, 256))
                else:
                    cur = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
                if POISON:
                    cur.fill_(255)
            self._ws[key] = cur
        if self.allocator is not None and cur.numel() != max(nbytes, 256):  # (hook mode: exactly the size asked for, every time)
            with torch.cuda.stream(self.stream):
                cur = self._ws[key] = self.allocator(max(nbytes, 256))
        return cur

    def synchronize(self):
        for st in reversed(self.streams):
            st.synchronize()

    def upload(self, dst: torch.Tensor, src_cpu: torch.Tensor):
        """H2D on the kernel stream.  Asynchronous only from PINNED memory (the caller keeps that buffer alive, e.g. the
        engine's frame staging); from pageable memory the call returns when the bytes have left the source, so a
        temporary may be passed (an async copy from a temporary that is freed right after read freed memory: seen as
        a garbage reference image on the MI355X)."""
        with torch.cuda.stream(self.stream):
            dst.copy_(src_cpu.view(dst.shape), non_blocking=src_cpu.is_pinned())

    def download(self, src: torch.Tensor) -> torch.Tensor:
        with torch.cuda.stream(self.stream):
            out = src.to("cpu", non_blocking=False)
        return out

    def download_into(self, dst_cpu: torch.Tensor, src: torch.Tensor):
        """D2H into a (pinned) host tensor on the kernel stream, then wait for that stream."""
        with torch.cuda.stream(self.stream):
            dst_cpu.view(src.shape).copy_(src, non_blocking=True)
        self.stream.synchronize()

    # ------------------------------------------------------------------ ops
    def conv(self, src0, src1, g: Geom, w: PackedConv, out, *, ldo=None, c0=None, c1=0, rowvec=None, residual=None,
             residual2=None, ldr=None, out_scale=1.0, act=L.ACT_NONE, out2=None, add2=None, out_t=None, ldt=0,
             t_col0=0, tile=None, split_k=None, workspace=None, pipeline=None, rowstat_out=None, ln_part=None,
             ln_eps=1e-5, chanstat_out=None, t_img=0, out_scale_dev=None, softmax_cols=0, _desc_only=False):
        """t_img: with g.batch > 1, columns of out_t per image (image b's pixels start at column b * t_img).
        out_scale_dev: one fp32 in device memory that replaces out_scale at run time (changeable under a captured graph)."""
        m = g.m
        c0 = c0 if c0 is not None else (w.cin - c1)
        if w.geglu:
            act = L.ACT_GEGLU
        key = self.conv_key(g, w, t_col0, self.epilogue_class(w, rowstat_out, ln_part, chanstat_out, residual2, out2, out_scale_dev, out_scale, act))
        inkernel = self.inkernel_splitk
        if tile is None:
            if key not in self.tile_override and key[-1] == 1 and key[:-1] + (0,) in self.tile_override:
                key = key[:-1] + (0,)  # (no throughput-mode entry for this shape: the alone-timed choice)
            if key in self.tile_override:
                # (the key names the epilogue class since round 4: an entry always fits the call it is looked up for -- rounds
                #  2-3 keyed on the shape alone and widened 64-column tiles / left the halo form after the fact)
                tile, split_k, inkernel, pipeline = self.tile_override[key]
            else:
                tile, sk = choose_tile(m, w.n, w.kp, w.geglu or w.tile128, t_col0 if out_t is not None else 0)
                split_k = sk if split_k is None else split_k
        split_k = split_k or 1
        if w.tile128:
            inkernel = True  # (the tile softmax runs in the reducing workgroup's epilogue)
        if pipeline == 10 and (self.no_c64 or out_scale_dev is not None or
                               not self._c64_call_ok(g, w, c1, act, out_scale, residual, residual2, out2, out_t, rowstat_out, chanstat_out, ln_part,
                                                     ldo if ldo is not None else w.n_out)):
            # (safety net, as below: an entry shared by a call the persistent form cannot take)
            pipeline, tile, split_k = (7, L.TILE_256x64, 1) if w.n % 8 == 0 else (3, L.TILE_64x64, 1)
        if pipeline == 7 and (self.no_halo or out_scale_dev is not None or
                              not self._halo_call_ok(g, w, c1, act, out_scale, residual2, out2, out_t, rowstat_out,
                                                     chanstat_out, ln_part)):
            # (safety net: a plain-class entry shared by a call the halo form cannot take -- e.g. SiLU after the residual)
            pipeline = 3
            tile = {L.TILE_256x128: L.TILE_128x128, L.TILE_256x64: L.TILE_128x64}.get(tile, tile)
            inkernel = True
        d = L.ConvDesc()
        d.src0, d.src1 = self._p(src0), self._p(src1)
        d.c0, d.c1 = c0, c1
        d.hs, d.ws, d.hi, d.wi, d.ho, d.wo = g.hs, g.ws, g.hi, g.wi, g.ho, g.wo
        d.ksize, d.stride, d.pad = g.ksize, g.stride, g.pad
        d.weight = self._p(w.weight)
        d.n, d.k, d.kp = w.n, w.k, w.kp
        d.bias = self._p(w.bias)
        if ln_part is not None:
            if w.ln_s is None:
                raise RuntimeError("conv: ln_part given but the weights were not packed with pack_linear_ln")
            d.ln_part, d.ln_groups, d.ln_eps = self._p(ln_part), ln_part.shape[1], ln_eps
            d.ln_s, d.ln_t = self._p(w.ln_s), self._p(w.ln_t)
        d.rowstat_out = self._p(rowstat_out)
        if chanstat_out is not None:
            d.chanstat_out = self._p(chanstat_out)
            d.chanstat_part = self._p(self.workspace("chanpart", (-(-m // 64)) * w.n * 8))
            d.chan_counters = self._p(self._counter_set(self._chan_counters, 4096))
        d.rowvec = self._p(rowvec)
        d.residual, d.residual2 = self._p(residual), self._p(residual2)
        d.ldr = ldr if ldr is not None else w.n_out
        d.out_scale = out_scale
        d.out_scale_dev = self._p(out_scale_dev)
        d.softmax_cols = softmax_cols
        d.act = act
        d.out = self._p(out)
        d.ldo = ldo if ldo is not None else w.n_out
        d.out2, d.add2 = self._p(out2), self._p(add2)
        d.out_t, d.ldt, d.t_col0 = self._p(out_t), ldt, t_col0
        d.batch, d.t_img = g.batch, t_img
        d.tile, d.split_k = tile, split_k
        d.pipeline = self.default_pipeline if pipeline is None else pipeline
        if split_k > 1:
            ws = workspace if workspace is not None else self.workspace("splitk", split_k * m * w.n * 4)
            d.workspace = self._p(ws)
            if inkernel:
                d.counters = self._p(self._counter_set(self._counters, L.SPLITK_MAX_TILES))
        elif workspace is not None:  # (development probes pass a buffer through)
            d.workspace = self._p(workspace)
        if _desc_only:
            return d
        self.ctx.call("vsd_conv_gemm", C.byref(d), self.s)

    # ---- several independent convs as ONE launch (include/vsd.h vsd_conv_gemm_group)
    GROUP_FORMS = [(L.TILE_64x64, 3), (L.TILE_64x64, 5), (L.TILE_64x128, 3), (L.TILE_64x128, 5), (L.TILE_128x64, 3), (L.TILE_128x64, 5),
                   (L.TILE_128x128, 3), (L.TILE_128x128, 5), (L.TILE_128x128, 8), (L.TILE_128x128, 9)]
    GROUP_ALONE = -1  # tile field of a group's table entry: "these members are faster as launches of their own"
    OWN_SPLIT = "own"  # conv_group(split=...): every member at the split ITS OWN table entry has (split field 0 in the group's entry)

    def own_splits(self, calls):
        """the split over K each member has as a launch of its own (its table entry / default form), or None when a member's own
        form sums over K in another order than the buffer-load ring does (the halo-patch form, the 256-row tiles' callers keep
        their form) -- then the group's bits would differ from the launches' and the members go out alone"""
        out = []
        for a, kw in calls:
            kw = {k: v for k, v in kw.items() if k not in ("tile", "split_k", "pipeline")}
            d = self.conv(*a, _desc_only=True, **kw)
            if d.pipeline == 7 or d.tile in (L.TILE_256x128, L.TILE_256x64, L.TILE_256x256):
                return None
            out.append(int(d.split_k))
        return out

    def group_key(self, calls, split=None):
        k = ["group"]
        for a, kw in calls:
            k += [a[2].m, a[3].n, a[3].kp]
        if split is not None:  # (a twin pair: the members' own split over K is part of the problem, see `pair`; "own": OWN_SPLIT)
            k += ["split", 0 if split == self.OWN_SPLIT else int(split)]
        return tuple(k) + (int(self.tune_mode),)

    def conv_group(self, calls, form=None, split=None, default=None):
        """calls: [(args, kwargs), ...] as for `conv` (1 x 1 / 3 x 3 layers on the buffer-load path): ONE launch for all of them
        (plus one reducer launch for the members that leave split-K slabs), every member in the same kernel form
        (tile, split_k, reduce in the launch, pipeline) -- `form`, else this group's entry of the tuning table (`tune_group`), else
        64 x 64 tiles on the 3-stage ring, unsplit (`split`: split that many times -- the table entry is then looked up for that
        split).  Member i works in a scratch set of its own (split-K slabs, counters).  Same bits as the members launched one by one at the
        same split_k, whatever their tile and pipeline (a conv's bits depend on the split alone: scripts/conv_bits_probe.py).  A
        table entry may also say that the members are better off alone.  split = OWN_SPLIT: members of DIFFERENT shapes that also
        exist as launches of their own in another form of the program (a ResnetBlock's conv1 and its shortcut conv) -- every member
        at the split its own table entry has (own_splits), the group's entry (split field 0) names tile / reduction form / pipeline."""
        if not 1 <= len(calls) <= L.CONV_GROUP_MAX:
            raise ValueError(f"conv_group: {len(calls)} members (1..{L.CONV_GROUP_MAX})")
        own = None
        if split == self.OWN_SPLIT:
            own = self.own_splits(calls)
            if own is None:
                form = (self.GROUP_ALONE, 1, True, 0)
        if form is None:
            ent = self.tile_override.get(self.group_key(calls, split))
            if ent is None and self.tune_mode == 1:
                ent = self.tile_override.get(self.group_key(calls, split)[:-1] + (0,))
            form = ent if ent is not None else (default or (L.TILE_64x64, (0 if own else split) or 1, True, 3))
        tile, split_k, inkernel, pipeline = form
        if tile == self.GROUP_ALONE:
            for a, kw in calls:
                self.conv(*a, **kw)
            return
        descs = (L.ConvDesc * len(calls))()
        saved = self.inkernel_splitk
        self.inkernel_splitk = bool(inkernel)
        try:
            for i, (a, kw) in enumerate(calls):
                kw = {k: v for k, v in kw.items() if k not in ("tile", "split_k", "pipeline")}
                # scratch set of member i on stream s: 2 i + s -- member 0 works in its stream's own set, the others in sets no launch of
                # EITHER stream uses (a group on the side stream beside split-K layers on the main one: the shortcut groups of the
                # two encoders.  With `i` alone the ControlNet's conv1 shared slabs and tickets with the UNet encoder's layers: wrong
                # frames from the captured two-stream form, found by the mini-pipeline parity tests)
                self._widx = 2 * i + self._sidx
                descs[i] = self.conv(*a, tile=tile, split_k=own[i] if own else split_k, pipeline=pipeline, _desc_only=True, **kw)
        finally:
            self._widx = None
            self.inkernel_splitk = saved
        self.ctx.call("vsd_conv_gemm_group", descs, len(calls), self.s)

    def group_candidates(self, calls, split=None):
        """the kernel forms `tune_group` times: GROUP_FORMS at the given split over K (slabs reduced by one more launch for the group,
        or in the launch), and the members as launches of their own"""
        stats = any(kw.get("rowstat_out") is not None or kw.get("chanstat_out") is not None or a[3].tile128 for a, kw in calls)
        if split == self.OWN_SPLIT:  # (entry's split field 0 = every member at its own)
            own = self.own_splits(calls)
            if own is None:
                return [(self.GROUP_ALONE, 1, True, 0)]
            sp, any_split = 0, max(own) > 1
        else:
            sp = split or 1
            any_split = sp > 1
        cands = [(t, sp, True, pl) for t, pl in self.GROUP_FORMS]
        if any_split and not stats:
            cands += [(t, sp, False, pl) for t, pl in self.GROUP_FORMS]
        return cands + [(self.GROUP_ALONE, 1, True, 0)]

    def tune_group(self, calls, reps: int = 12, split=None):
        """time the forms a group may take (`group_candidates`) back to back on this stream; remember the fastest"""
        table = []
        for form in self.group_candidates(calls, split):
            try:
                for _ in range(2):
                    self.conv_group(calls, form=form, split=split if split == self.OWN_SPLIT else None)
                best = 1e30
                for _trial in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                    for _ in range(reps):
                        self.conv_group(calls, form=form, split=split if split == self.OWN_SPLIT else None)
                    e1.record(self.stream)
                    e1.synchronize()
                    best = min(best, e0.elapsed_time(e1) / reps * 1e3)
                table.append((best,) + tuple(form))
            except RuntimeError:
                continue
        if not table:
            raise RuntimeError("tune_group: no kernel form ran for this group")
        table.sort()
        b = table[0]
        self.tile_override[self.group_key(calls, split)] = (b[1], b[2], bool(b[3]), b[4])
        return b, table

    # ---- two independent calls of one op as ONE grid per kernel (include/vsd.h vsd_pair_begin): the twin layers of the UNet and
    #      the ControlNet encoder.  a, b: recorded calls (bound method, args, kwargs).  b works in scratch set 1.
    def pair(self, a, b):
        (fa, aa, ka), (fb, ab, kb) = a, b
        if fa.__name__ == "conv" and fb.__name__ == "conv":
            sp = self.pair_split(aa, ka, ab, kb)
            if sp is not None:
                # Latency mode (one-frame programs): the pair's table entry (`tune_group`, timed alone), else the form its first member
                # has alone.  Throughput mode (coalesced launches on busy lanes): ALWAYS the members' own form -- a group's entry is
                # chosen by launch latency, and in such a form a pair of the 5-frame program's layers occupied 14 % more workgroup-time
                # than the two launches (profiles/round5f_pairs_at_5x4.txt); in their own form the pair costs what they cost and
                # saves the launch.
                own_form = self._pair_default if self.tune_mode == 1 else None
                return self.conv_group([(aa, ka), (ab, kb)], form=own_form, split=sp, default=self._pair_default)
            fa(*aa, **ka)
            self._widx = 1
            try:
                fb(*ab, **kb)
            finally:
                self._widx = None
            return
        self.ctx.call("vsd_pair_begin")
        done = False
        try:
            fa(*aa, **ka)
            self.ctx.call("vsd_pair_join")
            self._widx = 1
            fb(*ab, **kb)
            done = True
        finally:
            self._widx = None
            rc = self.ctx.lib.vsd_pair_end(self.ctx.h, None)  # (always closes the pair; held launches go out alone)
            if done:
                self.ctx.check(rc, "vsd_pair_end")

    def pair_split(self, aa, ka, ab, kb):
        """Two twin convs share a grid when both are on the buffer-load operand path (what vsd_conv_gemm_group takes) and the forms
        they have as launches of their own sum over K in the SAME order -- the same split_k, neither in the halo-patch form (its K
        order is channel block outer, tap inner).  Then the pair at that split gives the bits the two launches give, and a frame
        does not depend on which form of the program ran it.  -> that split_k, or None (launch them one by one)."""
        sp = None
        for a, k in ((aa, ka), (ab, kb)):
            g, w = a[2], a[3]
            if w.cin % 64 or (k.get("c1", 0) or 0) % 64 or (g.hi, g.wi) != (g.hs, g.ws) or g.ksize * g.ksize > 32:
                return None
            d = self.conv(*a, _desc_only=True, **k)
            if d.pipeline == 7 or d.tile in (L.TILE_256x128, L.TILE_256x64, L.TILE_256x256) or (sp is not None and d.split_k != sp):
                return None
            if sp is None:
                self._pair_default = (int(d.tile), int(d.split_k), bool(d.counters) or d.split_k <= 1, int(d.pipeline) if d.pipeline in (5, 8, 9) else 3)
            sp = int(d.split_k)
        return sp

    @staticmethod
    def _halo_call_ok(g, w, c1, act, out_scale, residual2, out2, out_t, rowstat_out, chanstat_out, ln_part) -> bool:
        """What vsd_conv_gemm's halo-patch form (pipeline 7) accepts: a 3x3 stride-1 conv with the plain epilogue."""
        plain = all(x is None for x in (residual2, out2, out_t, rowstat_out, chanstat_out, ln_part))
        return (plain and not w.geglu and g.ksize == 3 and g.stride == 1 and w.cin % 64 == 0 and (c1 or 0) % 64 == 0 and
                w.n % 8 == 0 and out_scale == 1.0 and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_SILU, L.ACT_RELU | L.ACT_POST))

    @classmethod
    def _c64_call_ok(cls, g, w, c1, act, out_scale, residual, residual2, out2, out_t, rowstat_out, chanstat_out, ln_part, ldo) -> bool:
        """What the persistent 64-input-channel form (pipeline 10, csrc/conv_c64.hip) accepts: the halo form's layer with Cin = 64 from one
        source and Cout = 64 -- or Cout <= 8 into 8-wide rows without a residual (TAESD's 64 -> 3 / 64 -> 4 projections)."""
        if w.cin != 64 or c1:
            return False
        if w.n == 64:
            return cls._halo_call_ok(g, w, c1, act, out_scale, residual2, out2, out_t, rowstat_out, chanstat_out, ln_part)
        plain = all(x is None for x in (residual, residual2, out2, out_t, rowstat_out, chanstat_out, ln_part))
        return (w.n <= 8 and ldo == 8 and plain and not w.geglu and g.ksize == 3 and g.stride == 1 and out_scale == 1.0 and
                act in (L.ACT_NONE, L.ACT_RELU, L.ACT_SILU))

    # Epilogue classes of the tuning key.  By scripts/wg_timeline.py the epilogue kind moves a workgroup's fixed cost from 1.0
    # to 4.6 us, and some classes exclude kernel forms (statistics outputs: no reducer kernel; softmax: 128-column tiles;
    # anything but the plain one: no halo-patch form) -- layers of one (M, N, K) with different epilogues get their own entry.
    EPI_PLAIN, EPI_LN, EPI_ROWSTAT, EPI_LN_ROWSTAT, EPI_SOFTMAX, EPI_GENERAL, EPI_PLAIN_ACT = range(7)

    @classmethod
    def epilogue_class(cls, w: PackedConv, rowstat_out=None, ln_part=None, chanstat_out=None, residual2=None, out2=None,
                       out_scale_dev=None, out_scale=1.0, act=0) -> int:
        if w.tile128 or (act & 0xff) == L.ACT_SOFTMAX:
            return cls.EPI_SOFTMAX
        if chanstat_out is not None or residual2 is not None or out2 is not None or out_scale_dev is not None or out_scale != 1.0:
            return cls.EPI_GENERAL
        if ln_part is not None:
            return cls.EPI_LN_ROWSTAT if rowstat_out is not None else cls.EPI_LN
        if rowstat_out is not None:
            return cls.EPI_ROWSTAT
        # (activations other than none / ReLU / SiLU / ReLU-after-residual leave the halo-patch form: quick-GELU of CLIP)
        return cls.EPI_PLAIN if (act & 0xff) in (L.ACT_NONE, L.ACT_RELU, L.ACT_SILU, L.ACT_GEGLU) else cls.EPI_PLAIN_ACT

    def conv_key(self, g: Geom, w: PackedConv, t_col0: int = 0, epi: int = 0):
        # last field: 0 = the choice that is fastest ALONE on an idle GPU (a lone launch: latency), 1 = the choice that costs the
        # least when FOUR launch lanes are busy (throughput): see tune_conv
        return (g.m, w.n, w.kp, g.ksize, g.stride, g.hi != g.hs or g.wi != g.ws, w.geglu, t_col0, int(epi), int(self.tune_mode))

    def conv_key_of(self, g: Geom, w: PackedConv, kwargs: dict):
        """the tuning key of a recorded conv call (args[2], args[3], its keyword arguments)"""
        return self.conv_key(g, w, kwargs.get("t_col0", 0), self.epilogue_class(
            w, kwargs.get("rowstat_out"), kwargs.get("ln_part"), kwargs.get("chanstat_out"), kwargs.get("residual2"), kwargs.get("out2"),
            kwargs.get("out_scale_dev"), kwargs.get("out_scale", 1.0), kwargs.get("act", L.ACT_NONE)))

    def tune_conv(self, args, kwargs, reps: int = 12):
        """Time every (tile, split_k, reduction form) candidate for one recorded conv call on the GPU and
        remember the fastest in tile_override.  Returns (best, table)."""
        g, w = args[2], args[3]
        t_col0 = kwargs.get("t_col0", 0)
        key = self.conv_key_of(g, w, kwargs)
        kt = w.kp // 64
        wide = w.geglu or w.tile128  # epilogues that need whole 128-column tiles and no split-K
        tiles = [L.TILE_128x128, L.TILE_64x128] if wide else [L.TILE_128x128, L.TILE_128x64, L.TILE_64x128, L.TILE_64x64]
        # the 256x128 tile (buffer-load path only: Cin % 64 == 0, no resize) pays when M is large (batched frames, TAESD)
        big_path = w.cin % 64 == 0 and (kwargs.get("c1", 0) or 0) % 64 == 0 and (g.hi, g.wi) == (g.hs, g.ws)
        big_ok = big_path and g.m >= 1024
        if big_ok:
            tiles = tiles + [L.TILE_256x128]
        # the 256x256 tile (eight waves, unsplit, row-banded epilogue): least L2 -> LDS fill per FLOP; for layers with enough of both
        # M and N to give the chip's lanes something to do (a softmax / transposed-output tile is 128 columns: not for those)
        huge_ok = (big_path and g.m >= 1024 and w.n >= 512 and not w.tile128 and kwargs.get("out_t") is None and
                   kwargs.get("chanstat_out") is None and not self.no_w8 and self.tune_mode == 1)
        # (both only among the throughput-mode candidates: alone on an idle chip the eight-wave forms measure 0-40 % slower than
        #  the four-wave ones, with four lanes busy 7-18 % faster on the wide 1 x 1 layers -- profiles/round6_w8_probe_*.txt)

        plain_epi = all(kwargs.get(k) is None for k in ("out2", "residual2", "out_t", "rowstat_out", "chanstat_out", "ln_part"))
        act = kwargs.get("act", L.ACT_NONE)
        halo_ok = (not wide and g.ksize == 3 and g.stride == 1 and w.cin % 64 == 0 and
                   (kwargs.get("c1", 0) or 0) % 64 == 0 and w.n % 8 == 0 and plain_epi and
                   kwargs.get("out_scale", 1.0) == 1.0 and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_SILU, L.ACT_RELU | L.ACT_POST))
        cands = [(L.TILE_256x256, 1, False, pl) for pl in (8, 9)] if huge_ok else []
        for t in tiles:
            bm, bn = L.TILE_DIMS[t]
            if kwargs.get("out_t") is not None and t_col0 % bn:
                continue
            blocks = -(-g.m // bm) * -(-w.n // bn)
            # pipelines 8 / 9: the 3-stage ring on eight waves (two per SIMD), buffer-load path, tiles of 128 x 128 and larger
            w8 = (8, 9) if big_path and bm * bn >= 128 * 128 and not self.no_w8 and self.tune_mode == 1 else ()
            for pl in ((3, 5) if t == L.TILE_256x128 else (0, 3, 4, 5, 6)) + w8:
                cands.append((t, 1, False, pl))
            if halo_ok and bm == 128:  # LDS halo patch (pipeline 7); split-K (over channel blocks) = the two-kernel form
                hblocks = g.batch * -(-g.ho // 8) * -(-g.wo // 16) * -(-w.n // bn)
                for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
                    if sp > w.cin // 64 or (sp > 1 and hblocks * sp > 1536):
                        break
                    cands.append((t, sp, False, 7))
                    if sp > 1 and hblocks <= L.SPLITK_MAX_TILES:
                        cands.append((t, sp, True, 7))
            if w.geglu or blocks >= 384:
                continue
            for sp in (2, 3, 4, 6, 8, 12, 16, 24):
                if sp > kt // 2 or blocks * sp > 1536:
                    break
                for pl in ((3, 5) if t == L.TILE_256x128 else (0, 3, 5)) + w8:
                    if kwargs.get("rowstat_out") is None and kwargs.get("chanstat_out") is None and not w.tile128:
                        cands.append((t, sp, False, pl))
                    cands.append((t, sp, True, pl))
        if halo_ok:  # 16x16 patches (8 waves): half the weight traffic of the 8x16 patch
            for t in (L.TILE_256x128, L.TILE_256x64):
                bn = L.TILE_DIMS[t][1]
                hblocks = g.batch * -(-g.ho // 16) * -(-g.wo // 16) * -(-w.n // bn)
                for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
                    if sp > w.cin // 64 or (sp > 1 and hblocks * sp > 1536):
                        break
                    cands.append((t, sp, False, 7))
                    if sp > 1 and hblocks <= L.SPLITK_MAX_TILES:
                        cands.append((t, sp, True, 7))
        # the persistent 64 -> 64 channel form (csrc/conv_c64.hip, pipeline 10): TAESD's block convs
        if not self.no_c64 and not wide and kwargs.get("out_scale_dev") is None and self._c64_call_ok(
                g, w, kwargs.get("c1", 0) or 0, act, kwargs.get("out_scale", 1.0), kwargs.get("residual"), kwargs.get("residual2"), kwargs.get("out2"),
                kwargs.get("out_t"), kwargs.get("rowstat_out"), kwargs.get("chanstat_out"), kwargs.get("ln_part"),
                kwargs.get("ldo") if kwargs.get("ldo") is not None else w.n_out):
            cands.append((L.TILE_256x64, 1, False, 10))
        kw = {k: v for k, v in kwargs.items() if k not in ("tile", "split_k", "pipeline")}
        table = []
        saved = self.tile_override.pop(key, None)
        if self.tune_mode == 1:
            # Throughput mode.  Timed alone, a candidate is judged by its launch latency on an idle chip -- small tiles and many
            # workgroups win.  With four lanes busy the chip is shared, other launches fill whatever a launch leaves idle, and
            # what a candidate costs is its CU-time: larger tiles (less operand re-streaming per FLOP) win back.  Measured on
            # MI355X (scripts/tune_lanes_probe.py, us per launch alone | with four copies in flight): 5x32x32 640->640 3x3:
            # 64x128 GEMM form 54.8 | 48.9, 256x128 halo form 81.6 | 38.7; 1280x1280x5120: 64x64 35.2 | 30.4, 128x64 42.3 | 23.8.
            # So: every candidate alone first (cheap), then the ones within 2.5x of the best with four copies at once.
            alone = self._time_candidates(args, kw, cands, reps)
            if not alone:
                raise RuntimeError("tune_conv: no candidate ran")
            cut = 2.5 * alone[0][0]
            # (the 256 x 256 tile always makes the shortlist: a handful of workgroups is slow ALONE by construction -- 2-5 x on an idle
            #  chip -- and what it is for is the shared one)
            short = [(t, sp, ink, pl) for (us, t, sp, ink, pl) in alone if us <= cut or (t == L.TILE_256x256 and us <= 6.0 * alone[0][0])]
            table = self._time_candidates_on_lanes(args, kw, short, reps) or alone
            self.inkernel_splitk = True
            best = table[0]
            self.tile_override[key] = (best[1], best[2], best[3], best[4])
            return best, table
        last_error = None
        for (t, sp, ink, pl) in cands:
            self.inkernel_splitk = ink
            try:
                for _ in range(2):
                    self.conv(*args, tile=t, split_k=sp, pipeline=pl, **kw)
                best_us = 1e30
                for _trial in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.stream)
                    for _ in range(reps):
                        self.conv(*args, tile=t, split_k=sp, pipeline=pl, **kw)
                    e1.record(self.stream)
                    e1.synchronize()
                    best_us = min(best_us, e0.elapsed_time(e1) / reps * 1e3)
                table.append((best_us, t, sp, ink, pl))
            except RuntimeError as e:
                last_error = e
                continue
        self.inkernel_splitk = True
        if not table:  # (every candidate was refused: say for which layer and why instead of an IndexError)
            raise RuntimeError(f"tune_conv: no candidate ran for M={g.m} N={w.n} K={w.k} ksize={g.ksize}: {last_error}")
        table.sort()
        best = table[0]
        if best[2] > 1 and not best[3] and self.one_launch_bias_us > 0:
            # a two-launch form won: take a one-launch form that is within the bias instead (one graph node less; back-to-back
            # timing on an otherwise idle GPU flatters the second launch, which in a frame also delays the NEXT layer's start)
            for cand in table[1:]:
                if cand[0] > best[0] + self.one_launch_bias_us:
                    break
                if cand[2] == 1 or cand[3]:
                    best = cand
                    break
        self.tile_override[key] = (best[1], best[2], best[3], best[4])
        return best, table

    def _time_candidates(self, args, kw, cands, reps):
        """us per launch of every candidate, alone, back to back on this object's stream; sorted"""
        table = []
        for (t, sp, ink, pl) in cands:
            self.inkernel_splitk = ink
            try:
                for _ in range(2):
                    self.conv(*args, tile=t, split_k=sp, pipeline=pl, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
                for _ in range(reps):
                    self.conv(*args, tile=t, split_k=sp, pipeline=pl, **kw)
                e1.record(self.stream)
                e1.synchronize()
                table.append((e0.elapsed_time(e1) / reps * 1e3, t, sp, ink, pl))
            except RuntimeError:
                continue
        self.inkernel_splitk = True
        table.sort()
        return table

    def _time_candidates_on_lanes(self, args, kw, cands, reps, n_lanes: int = 4):
        """us per launch with `n_lanes` copies of the candidate in flight, one per launch lane (captured graphs of `reps`
        launches each, so that the host is out of the picture); every lane has its own input / output / scratch"""
        # this object on ITS lane, helpers on the other lanes (one set per process, kept on the shared lane book; an engine on
        # lane 1-3 that tunes online must not put two copies on its own stream: ADVICE r4)
        others = [l for l in range(L.POOL_STREAMS) if l != self.lane % L.POOL_STREAMS][:max(0, n_lanes - 1)]
        for l in others:
            if l not in self._lanes.tune_helpers:
                self._lanes.tune_helpers[l] = self.clone(lane=l, _register=False)
        lanes = [self] + [self._lanes.tune_helpers[l] for l in others]
        for o in lanes:
            o.tune_mode = self.tune_mode
        src0, src1, g, w, out = args[:5]
        # every lane writes its OWN buffers: the output and every other tensor the launch writes (transposed output, second
        # output, statistics) -- four launches at once into one buffer would time a write conflict, not the candidate
        writes = ("out2", "out_t", "rowstat_out", "chanstat_out")
        per_lane = [(src0, src1, out, kw)]
        for _ in lanes[1:]:
            kwl = dict(kw)
            for name in writes:
                if kwl.get(name) is not None:
                    kwl[name] = kwl[name].clone()
            per_lane.append((src0.clone(), None if src1 is None else src1.clone(), out.clone(), kwl))
        table = []
        for (t, sp, ink, pl) in cands:
            graphs = []
            try:
                for o, (a0, a1, oo, kwl) in zip(lanes, per_lane):
                    o.inkernel_splitk = ink
                    o.conv(a0, a1, g, w, oo, *args[5:], tile=t, split_k=sp, pipeline=pl, **kwl)
                    o.synchronize()
                    o.graph_begin()
                    try:
                        for _ in range(reps):
                            o.conv(a0, a1, g, w, oo, *args[5:], tile=t, split_k=sp, pipeline=pl, **kwl)
                    finally:
                        graphs.append((o, o.graph_end()))
                best = 1e30
                for _trial in range(3):
                    for o, gr in graphs:
                        o.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        for o, gr in graphs:
                            o.graph_launch(gr)
                    for o, gr in graphs:
                        o.synchronize()
                    best = min(best, (time.perf_counter() - t0) / (2 * reps * len(graphs)) * 1e6)
                table.append((best, t, sp, ink, pl))
            except RuntimeError:
                pass
            finally:
                for o, gr in graphs:
                    o.graph_destroy(gr)
                for o in lanes:
                    o.inkernel_splitk = True
        table.sort()
        return table

    # ---- tuning table persistence (the "find" results are per device generation and shape)
    def save_tuning(self, path: str):
        """Write this process's choices INTO the table at `path` (entries of other shapes -- other batch sizes, other
        frame sizes -- are kept)."""
        import json
        import os

        merged = {}
        if os.path.exists(path):
            try:
                for k, v in json.load(open(path)).get("table", []):
                    merged[tuple(bool(x) if isinstance(x, bool) else x for x in k)] = tuple(v)
            except Exception:
                merged = {}
        merged.update(self.tile_override)
        with open(path, "w") as f:
            json.dump({"device": torch.cuda.get_device_name(self.device),
                       "table": [[list(k), list(v)] for k, v in sorted(merged.items(), key=str)]}, f, indent=0)

    def load_tuning(self, path: str) -> int:
        import json
        import os

        if not os.path.exists(path):
            return 0
        d = json.load(open(path))
        n = 0
        for k, v in d.get("table", []):
            key = tuple(bool(x) if isinstance(x, bool) else x for x in k)
            if isinstance(key[-1], bool):
                continue  # a round-3 table (last field: "has a statistics output"): its entries name no epilogue class
            if len(key) == 9 and key[0] != "group":
                key = key + (0,)  # (a table written before the tuning mode joined the key: timed alone)
            if key not in self.tile_override:
                self.tile_override[key] = (int(v[0]), int(v[1]), bool(v[2]), int(v[3]))
                n += 1
        return n

    # ---- fused per-token chains of a transformer block at the 320-wide level (csrc/fused_tail.hip)
    TAIL_C = 320

    def tail_a(self, att, h, m, out1: PackedConv, q2: PackedConv, h1_out, q_out, ln_eps=1e-5):
        """h1 = att Wo^T + b + h ; q = LN(h1) Wq'^T"""
        self.ctx.call("vsd_tail_a", self._p(att), self._p(h), m, self._p(out1.weight_frag), self._p(out1.bias), self._p(q2.weight_frag),
                      self._p(q2.ln_s), self._p(q2.ln_t), ln_eps, self._p(h1_out), self._p(q_out), self.s)

    def tail_b(self, att2, h1, x, m, out2: PackedConv, ff1: PackedConv, ff2: PackedConv, proj: PackedConv, out, ln_eps=1e-5):
        """h2 = att2 Wo^T + b + h1 ; h3 = GEGLU-FF(LN(h2)) + h2 ; out = h3 Wp^T + b + x"""
        self.ctx.call("vsd_tail_b", self._p(att2), self._p(h1), self._p(x), m, self._p(out2.weight_frag), self._p(out2.bias),
                      self._p(ff1.weight_frag), self._p(ff1.ln_s), self._p(ff1.ln_t), ln_eps, self._p(ff2.weight_frag),
                      self._p(ff2.bias), self._p(proj.weight_frag), self._p(proj.bias), self._p(out), self.s)

    def groupnorm_launches(self, c0, c1, hw, groups, batch=1) -> int:
        """kernel launches one `groupnorm` call issues (the library's own decision: csrc/norm.hip)"""
        return int(self.ctx.lib.vsd_groupnorm_launches(c0, c1, hw, max(1, batch), groups))

    def groupnorm(self, src0, src1, c0, c1, hw, groups, eps, gamma, beta, silu, out, batch=1):
        ws = self.workspace("gn", max(1, batch) * int(self.ctx.lib.vsd_groupnorm_workspace_bytes(hw, c0 + c1, groups)))
        if batch > 1:
            self.ctx.call("vsd_groupnorm_batched", self._p(src0), self._p(src1), c0, c1, hw, batch, groups, eps,
                          self._p(gamma), self._p(beta), int(silu), self._p(out), self._p(ws), self.s)
            return
        self.ctx.call("vsd_groupnorm", self._p(src0), self._p(src1), c0, c1, hw, groups, eps, self._p(gamma),
                      self._p(beta), int(silu), self._p(out), self._p(ws), self.s)

    def layernorm(self, x, rows, c, gamma, beta, eps, out):
        self.ctx.call("vsd_layernorm", self._p(x), rows, c, self._p(gamma), self._p(beta), eps, self._p(out), self.s)

    def attention(self, q, ldq, k, ldk, vt, ldvt, out, ldo, sq, sk, heads, d, scale, causal=False, batch=1, k_brows=0,
                  vt_bcols=0):
        """batch > 1: image b uses q/out rows [b*sq, (b+1)*sq), K rows from b*k_brows, V^T columns from b*vt_bcols."""
        if batch > 1:
            self.ctx.call("vsd_attention_batched", self._p(q), ldq, self._p(k), ldk, self._p(vt), ldvt, self._p(out), ldo,
                          sq, sk, heads, d, scale, int(causal), batch, k_brows, vt_bcols, self.s)
            return
        self.ctx.call("vsd_attention", self._p(q), ldq, self._p(k), ldk, self._p(vt), ldvt, self._p(out), ldo, sq, sk,
                      heads, d, scale, int(causal), self.s)

    def preprocess_rgb(self, rgb_u8, h, w, out):
        self.ctx.call("vsd_preprocess_rgb", self._p(rgb_u8), h, w, self._p(out), self.s)

    def sobel_control(self, rgb_u8, h, w, low, high, edge_u8, control_out):
        ws = self.workspace("sobel", int(self.ctx.lib.vsd_sobel_workspace_bytes(h, w)))
        self.ctx.call("vsd_sobel_control", self._p(rgb_u8), h, w, low, high, self._p(edge_u8), self._p(control_out),
                      self._p(ws), self.s)

    def add_noise(self, x0, noise_f32, sqrt_a, sqrt_b, hw, out):
        self.ctx.call("vsd_add_noise", self._p(x0), self._p(noise_f32), sqrt_a, sqrt_b, hw, self._p(out), self.s)

    def lcm_step(self, eps, sample, noise_f32, coef, hw, prev, denoised, dec_in=None):
        arr = (C.c_float * 6)(*[float(x) for x in coef])
        self.ctx.call("vsd_lcm_step", self._p(eps), self._p(sample), self._p(noise_f32), arr, hw, self._p(prev),
                      self._p(denoised), self._p(dec_in), self.s)

    def add_noise_dev(self, x0, noise_f32, coef_dev, hw, batch, out):
        """coef_dev: fp32 [2] in device memory (sqrt_a, sqrt_b); `batch` images per launch, one noise draw for all."""
        self.ctx.call("vsd_add_noise_dev", self._p(x0), self._p(noise_f32), self._p(coef_dev), hw, batch, self._p(out), self.s)

    def lcm_step_dev(self, eps, sample, noise_f32, coef_dev, hw, batch, prev, denoised, dec_in=None):
        """coef_dev: fp32 [6] in device memory, as `lcm_step`'s coef."""
        self.ctx.call("vsd_lcm_step_dev", self._p(eps), self._p(sample), self._p(noise_f32), self._p(coef_dev), hw, batch,
                      self._p(prev), self._p(denoised), self._p(dec_in), self.s)

    def adain(self, x, stats, stats_ref, rows, c, out, eps=1e-6):
        """reference-only AdaIN: per-channel re-normalisation of x to the banked statistics (fp32 [c][2] sum / sumsq)"""
        self.ctx.call("vsd_adain", self._p(x), self._p(stats), self._p(stats_ref), rows, c, eps, self._p(out), self.s)

    def xattn_fold(self, k, vt, tl, wq, wo, gamma, beta, c, heads, scale, xa1_w, xa1_s, xa1_t, xa2_w):
        """per-prompt constants of one absorbed cross-attention block (include/vsd.h vsd_xattn_fold)"""
        self.ctx.call("vsd_xattn_fold", self._p(k), k.stride(0), self._p(vt), vt.stride(0), tl, self._p(wq), self._p(wo), self._p(gamma),
                      self._p(beta), c, heads, float(scale), self._p(xa1_w), self._p(xa1_s), self._p(xa1_t), self._p(xa2_w), self.s)

    def embed_tokens(self, ids_i64, tok_emb, pos_emb, out):
        n, c = out.shape
        self.ctx.call("vsd_embed_tokens", self._p(ids_i64), self._p(tok_emb), self._p(pos_emb), n, c, tok_emb.shape[0],
                      self._p(out), self.s)

    def postprocess_rgb(self, img, ld, hw, rgb_u8):
        self.ctx.call("vsd_postprocess_rgb", self._p(img), ld, hw, self._p(rgb_u8), self.s)

    def axpy(self, a, b, scale, n, out):
        self.ctx.call("vsd_axpy", self._p(a), self._p(b), scale, n, self._p(out), self.s)

    # ------------------------------------------------------------------ graphs / profiling
    def graph_begin(self):
        self.ctx.call("vsd_graph_begin", self.s)

    def graph_end(self):
        g = C.c_void_p()
        self.ctx.call("vsd_graph_end", self.s, C.byref(g))
        return g

    def graph_launch(self, g):
        self.ctx.call("vsd_graph_launch", g, self.s)

    def graph_destroy(self, g):
        self.ctx.call("vsd_graph_destroy", g)

    # ---- launch sequences (include/vsd.h vsd_seq): single-branch graphs on this object's streams + event edges between them
    def seq_create(self):
        q = C.c_void_p()
        self.ctx.call("vsd_seq_create", C.byref(q))
        return q

    def seq_capture_begin(self, sidx: int):
        self._sidx = sidx
        self.ctx.call("vsd_graph_begin", self.raw_stream(sidx))

    def seq_capture_end(self, seq, sidx: int):
        g = C.c_void_p()
        self.ctx.call("vsd_graph_end", self.raw_stream(sidx), C.byref(g))
        self.ctx.call("vsd_seq_add_graph", seq, g, self.raw_stream(sidx))

    def seq_record(self, seq, sidx: int) -> int:
        ev = C.c_int()
        self.ctx.call("vsd_seq_add_record", seq, self.raw_stream(sidx), C.byref(ev))
        return int(ev.value)

    def seq_wait(self, seq, sidx: int, ev: int):
        self.ctx.call("vsd_seq_add_wait", seq, self.raw_stream(sidx), int(ev))

    def seq_count(self, seq):
        g, e = C.c_int(), C.c_int()
        self.ctx.call("vsd_seq_count", seq, C.byref(g), C.byref(e))
        return int(g.value), int(e.value)

    def seq_launch(self, seq):
        self.ctx.call("vsd_seq_launch", seq)

    def seq_destroy(self, seq):
        self.ctx.call("vsd_seq_destroy", seq)

    def profile_begin(self):
        self.ctx.call("vsd_profile_begin")

    def profile_overhead_ms(self, n: int = 200) -> float:
        ms = C.c_float()
        self.ctx.call("vsd_profile_overhead", self.s, n, C.byref(ms))
        return float(ms.value)

    def profile_end(self):
        self.ctx.call("vsd_profile_end")
        return self.ctx.stage_times()
