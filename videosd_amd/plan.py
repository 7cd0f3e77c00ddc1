"""A prepared frame program as a FILE that a host without Python can run (include/vsd.h vsd_plan_load / vsd_plan_infer).

SURVEY.md section 8b sketched whole-frame C entry points (vsd_load_weights / vsd_prepare / vsd_infer); the sequencing of a frame lives
in engine.py, so rounds 1-5 exported ops only and a non-Python host could not drive a frame (missing #3).  This closes
that without a second engine: `export_plan(engine, path)` runs the engine's one-stream program ONCE with every C-ABI call recorded
-- entry point, scalar arguments, descriptors -- and every device pointer rewritten as (region, offset): a region is one allocation
of the process (weights, per-plan constants, the prompt block, counters, I/O buffers: saved with their bytes; activations and
workspaces: size only).  `vsd_plan_load` allocates the regions, uploads the saved bytes, patches the pointers, replays the calls
under stream capture and keeps the graph; `vsd_plan_infer` is upload, launch, download.  Same kernels, same arguments: the frame a plan
produces is bit for bit the engine's (tests/test_plan_gpu.py).  A plan is one (size, steps, strength, ControlNet scale, prompt, frames
per launch); the reference's per-frame options (videopipeline.py:75-128) mean another plan.

File (little endian): "VSDPLAN1", u32 version, H, W, batch, n_regions, n_calls, u32 in_region, u64 in_offset, u32 out_region, u64
out_offset, u32 prompt_region, u64 prompt_offset, u64 prompt_bytes (the engine's prompt block: vsd_plan_load_prompt replaces its bytes
with another prompt's, `export_prompt`); regions: u64 size, u32 saved, u32 0; calls: u32 entry point (PLAN_FUNCS index), u32 nargs, args of 16 bytes
(u32 tag, u32 aux, u64 value): 0 int32, 1 float32 bits, 2 pointer (aux = region, value = offset), 3 null, 4 the plan's stream,
5 descriptor array (aux = count, value = bytes; followed by the bytes, u32 nfix, nfix x (u32 byte offset, u32 region, u64 offset));
then the bytes of the saved regions in order."""
import bisect
import ctypes as C
import struct

import numpy as np
import torch

from . import lib as L

MAGIC = b"VSDPLAN1"
VERSION = 1
# entry points a frame program may call, by id (csrc/plan_dispatch.inc is generated from this list: scripts/gen_plan_dispatch.py)
PLAN_FUNCS = ["vsd_preprocess_rgb", "vsd_sobel_control", "vsd_conv_gemm", "vsd_conv_gemm_group", "vsd_pair_begin", "vsd_pair_join",
              "vsd_pair_end", "vsd_groupnorm", "vsd_groupnorm_batched", "vsd_attention", "vsd_attention_batched", "vsd_tail_a", "vsd_tail_b",
              "vsd_add_noise_dev", "vsd_lcm_step_dev", "vsd_postprocess_rgb", "vsd_adain", "vsd_layernorm"]
T_I32, T_F32, T_PTR, T_NULL, T_STREAM, T_DESC = range(6)
_PTR_FIELDS = [(name, getattr(L.ConvDesc, name).offset) for name, t in L.ConvDesc._fields_ if t is C.c_void_p]


class _Regions:
    """the process's live device allocations (torch's caching allocator) that the program points into"""

    def __init__(self, device):
        blocks = []
        for seg in torch.cuda.memory_snapshot():
            if seg.get("device", 0) != device:
                continue
            addr = seg["address"]
            for b in seg["blocks"]:
                if b["state"].startswith("active"):
                    blocks.append((b.get("address", addr), b["size"]))
                addr += b["size"]
        blocks.sort()
        self.starts = [a for a, _ in blocks]
        self.blocks = blocks
        self.used = {}  # block index -> region id

    def locate(self, ptr: int):
        i = bisect.bisect_right(self.starts, ptr) - 1
        if i < 0 or ptr >= self.blocks[i][0] + self.blocks[i][1]:
            raise RuntimeError(f"plan export: device pointer {ptr:#x} lies in no live allocation of this process")
        rid = self.used.setdefault(i, len(self.used))
        return rid, ptr - self.blocks[i][0]


class _LibProxy:
    """the ctypes library with vsd_pair_end recorded (HipOps.pair calls it past Context.call)"""

    def __init__(self, lib, rec):
        self._lib, self._rec = lib, rec

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name != "vsd_pair_end":
            return fn

        def pair_end(h, joined):
            self._rec("vsd_pair_end", (None,))
            return fn(h, joined)

        return pair_end


def export_plan(engine, path: str) -> dict:
    """Write the prepared engine's program (one-stream form) to `path`.  The engine must have run a frame (prompt installed)."""
    from .engine import Engine

    ops = engine.ops
    if engine.plan is None:
        raise RuntimeError("export_plan: prepare the engine first")
    ops.synchronize()
    calls = []

    def rec(name, args):
        calls.append((name, args))

    ctx = ops.ctx
    orig_call, orig_lib = ctx.call, ctx.lib

    def recording_call(name, *args):
        rec(name, args)
        return orig_call(name, *args)

    ctx.call, ctx.lib = recording_call, _LibProxy(orig_lib, rec)
    try:
        if hasattr(ops, "tune_mode"):
            ops.tune_mode = 1 if engine.tune_for_lanes else 0
        with torch.cuda.stream(ops.stream):
            for fn, a, k in engine.program_serial.calls:
                if getattr(fn, "__name__", "") in Engine.SYNC_OPS:
                    continue
                fn(*a, **k)
        ops.synchronize()
    finally:
        ctx.call, ctx.lib = orig_call, orig_lib
    regs = _Regions(ops.device.index if hasattr(ops.device, "index") and ops.device.index is not None else 0)
    stream = ops.s.value
    if not stream:
        raise RuntimeError("plan export: the engine launches on the null stream")
    body = bytearray()
    for name, args in calls:
        if name not in PLAN_FUNCS:
            raise RuntimeError(f"plan export: the program calls {name}, which a plan cannot hold")
        types = L.SIGNATURES[name][1][1:]
        if len(types) != len(args):
            raise RuntimeError(f"plan export: {name} recorded with {len(args)} arguments, declared with {len(types)}")
        body += struct.pack("<II", PLAN_FUNCS.index(name), len(args))
        for t, v in zip(types, args):
            if t is C.c_void_p:
                v = v.value if isinstance(v, C.c_void_p) else v
                if v is None or v == 0:
                    body += struct.pack("<IIQ", T_NULL, 0, 0)
                elif int(v) == stream:
                    body += struct.pack("<IIQ", T_STREAM, 0, 0)
                else:
                    rid, off = regs.locate(int(v))
                    body += struct.pack("<IIQ", T_PTR, rid, off)
            elif t in (C.c_int, C.c_int32):
                body += struct.pack("<IIq", T_I32, 0, int(v))
            elif t is C.c_float:
                body += struct.pack("<IIQ", T_F32, 0, struct.unpack("<I", struct.pack("<f", float(v)))[0])
            elif t == C.POINTER(C.c_int):
                body += struct.pack("<IIQ", T_NULL, 0, 0)  # (vsd_pair_end's optional output)
            elif t == C.POINTER(L.ConvDesc):
                obj = v._obj if hasattr(v, "_obj") else v  # byref(desc) | an array of descriptors
                count = len(obj) if isinstance(obj, C.Array) else 1
                raw = C.string_at(C.addressof(obj), C.sizeof(obj))
                fix = []
                for j in range(count):
                    d = obj[j] if count > 1 or isinstance(obj, C.Array) else obj
                    for fname, foff in _PTR_FIELDS:
                        p = getattr(d, fname)
                        if p:
                            rid, off = regs.locate(int(p))
                            fix.append((j * C.sizeof(L.ConvDesc) + foff, rid, off))
                body += struct.pack("<IIQ", T_DESC, count, len(raw)) + raw + struct.pack("<I", len(fix))
                for boff, rid, off in fix:
                    body += struct.pack("<IIQ", boff, rid, off)
            else:
                raise RuntimeError(f"plan export: {name}: argument type {t} has no plan encoding")
    pr_r, pr_off = regs.locate(engine.pblock.buf.data_ptr())
    in_r, in_off = regs.locate(engine.frame_u8.data_ptr())
    out_r, out_off = regs.locate(engine.out_u8.data_ptr())
    # what is scratch (no bytes saved): the activation arena and the op workspaces
    scratch = {t.data_ptr() for t in engine.arena.chunks}
    scratch |= {t.data_ptr() for t in ops._ws.values()}
    order = sorted(regs.used.items(), key=lambda kv: kv[1])
    p = engine.plan
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<IIIIII", VERSION, p["H"], p["W"], p["batch"], len(order), len(calls)))
        f.write(struct.pack("<IQIQ", in_r, in_off, out_r, out_off))
        f.write(struct.pack("<IQQ", pr_r, pr_off, engine.pblock.buf.numel()))
        saved = []
        for bi, rid in order:
            addr, size = regs.blocks[bi]
            keep = not any(addr <= s < addr + size for s in scratch)
            saved.append(keep)
            f.write(struct.pack("<QII", size, int(keep), 0))
        f.write(body)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        for (bi, rid), keep in zip(order, saved):
            if not keep:
                continue
            addr, size = regs.blocks[bi]
            host = np.empty(size, dtype=np.uint8)
            rc = hip.hipMemcpy(host.ctypes.data, addr, size, 2)  # hipMemcpyDeviceToHost
            if rc != 0:
                raise RuntimeError(f"plan export: hipMemcpy of a {size}-byte region failed ({rc})")
            f.write(host.tobytes())
    return {"regions": len(order), "saved_bytes": sum(regs.blocks[bi][1] for (bi, _), k in zip(order, saved) if k),
            "scratch_bytes": sum(regs.blocks[bi][1] for (bi, _), k in zip(order, saved) if not k), "calls": len(calls)}


PROMPT_MAGIC = b"VSDPRMT1"


def export_prompt(pblock, path: str) -> int:
    """Another prompt's constants (an engine.PromptBlock: cross-attention K / V^T of every layer and the folded weights, built on the GPU
    from the text encoder's output) as a file for vsd_plan_load_prompt: a C host changes the prompt of a loaded plan without a new plan.
    The block must have the plan's layout (same networks, same text length)."""
    data = pblock.buf.cpu().numpy().tobytes()
    with open(path, "wb") as f:
        f.write(PROMPT_MAGIC + struct.pack("<Q", len(data)) + data)
    return len(data)


class CPlan:
    """vsd_plan_load / vsd_plan_infer through ctypes (tests, examples): what a C host does with a plan file"""

    def __init__(self, path: str, device_id: int = 0):
        self.ctx = L.Context(device_id)
        h = C.c_void_p()
        self.ctx.call("vsd_plan_load", path.encode(), C.byref(h))
        self.h = h
        dims = (C.c_int * 3)()
        self.ctx.call("vsd_plan_info", self.h, dims)
        self.H, self.W, self.batch = int(dims[0]), int(dims[1]), int(dims[2])

    def infer(self, frame: np.ndarray) -> np.ndarray:
        want = (self.H, self.W, 3) if self.batch == 1 else (self.batch, self.H, self.W, 3)
        if frame.shape != want or frame.dtype != np.uint8:
            raise ValueError(f"frame must be uint8 {want}")
        frame = np.ascontiguousarray(frame)
        out = np.empty_like(frame)
        self.ctx.call("vsd_plan_infer", self.h, frame.ctypes.data, out.ctypes.data)
        return out

    def load_prompt(self, path: str):
        self.ctx.call("vsd_plan_load_prompt", self.h, path.encode())

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.vsd_plan_free(self.ctx.h, self.h)
            self.h = None
