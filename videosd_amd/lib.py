"""ctypes binding of libvsd.so (include/vsd.h).  No fallback: if the HIP library is missing or a call
fails, a RuntimeError is raised."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VSD_LIB: another build of the same sources (development: the instrumented libvsd_tl.so of build.build_timeline)
LIB_PATH = os.environ.get("VSD_LIB") or os.path.join(HERE, "libvsd.so")

VERSION = 6  # include/vsd.h VSD_VERSION
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GEGLU, ACT_QUICKGELU, ACT_SOFTMAX, ACT_GELU = range(7)
ACT_POST = 256
SPLITK_MAX_TILES = 16384
POOL_STREAMS = 4  # include/vsd.h VSD_POOL_STREAMS
CONV_GROUP_MAX = 8  # include/vsd.h VSD_CONV_GROUP_MAX
TILE_128x128, TILE_128x64, TILE_64x64, TILE_64x128, TILE_256x128, TILE_256x64, TILE_256x256 = range(7)
TILE_DIMS = {TILE_128x128: (128, 128), TILE_128x64: (128, 64), TILE_64x64: (64, 64), TILE_64x128: (64, 128),
             TILE_256x128: (256, 128), TILE_256x64: (256, 64), TILE_256x256: (256, 256)}
FAMILIES = ["conv_gemm", "splitk_reduce", "groupnorm", "layernorm", "attention", "elementwise"]


class ConvDesc(C.Structure):
    _fields_ = [
        ("src0", C.c_void_p), ("src1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("hs", C.c_int32), ("ws", C.c_int32),
        ("hi", C.c_int32), ("wi", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("weight", C.c_void_p),
        ("n", C.c_int32), ("k", C.c_int32), ("kp", C.c_int32),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("residual", C.c_void_p), ("residual2", C.c_void_p),
        ("ldr", C.c_int32),
        ("out_scale", C.c_float),
        ("act", C.c_int32),
        ("out", C.c_void_p), ("ldo", C.c_int32),
        ("out2", C.c_void_p), ("add2", C.c_void_p),
        ("out_t", C.c_void_p), ("ldt", C.c_int32), ("t_col0", C.c_int32),
        ("tile", C.c_int32), ("split_k", C.c_int32),
        ("workspace", C.c_void_p),
        ("pipeline", C.c_int32),
        ("rowstat_out", C.c_void_p), ("chanstat_out", C.c_void_p), ("chanstat_part", C.c_void_p),
        ("chan_counters", C.c_void_p), ("ln_part", C.c_void_p), ("ln_groups", C.c_int32), ("ln_eps", C.c_float),
        ("ln_s", C.c_void_p), ("ln_t", C.c_void_p),
        ("counters", C.c_void_p),
        ("batch", C.c_int32), ("t_img", C.c_int32),
        ("out_scale_dev", C.c_void_p),
        ("softmax_cols", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol declared in include/vsd.h
SIGNATURES = {
    "vsd_version": (C.c_int, []),
    "vsd_conv_desc_size": (C.c_int, []),
    "vsd_xattn_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_create": (C.c_void_p, [C.c_int]),
    "vsd_destroy": (None, [C.c_void_p]),
    "vsd_last_error": (C.c_char_p, [C.c_void_p]),
    "vsd_conv_gemm": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc), C.c_void_p]),
    "vsd_conv_gemm_group": (C.c_int, [C.c_void_p, C.POINTER(ConvDesc), C.c_int, C.c_void_p]),
    "vsd_pair_begin": (C.c_int, [C.c_void_p]),
    "vsd_pair_join": (C.c_int, [C.c_void_p]),
    "vsd_pair_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "vsd_tail_a": (C.c_int, [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 3),
    "vsd_tail_b": (C.c_int, [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 6),
    "vsd_groupnorm_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "vsd_groupnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                C.c_void_p, C.c_void_p]),
    "vsd_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "vsd_attention_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]),
    "vsd_groupnorm_batched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_groupnorm_launches": (C.c_int, [C.c_int] * 5),
    "vsd_preprocess_rgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vsd_sobel_workspace_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "vsd_sobel_control": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_add_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                C.c_void_p]),
    "vsd_lcm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_add_noise_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vsd_lcm_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_adain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "vsd_embed_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vsd_postprocess_rgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vsd_axpy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p, C.c_void_p]),
    "vsd_graph_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vsd_graph_end": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "vsd_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_graph_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vsd_stream_pool": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "vsd_stream_pool_check": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "vsd_stream_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_void_p)]),
    "vsd_stream_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vsd_seq_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "vsd_seq_add_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_seq_add_record": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "vsd_seq_add_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "vsd_seq_count": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vsd_seq_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vsd_seq_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vsd_plan_load": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "vsd_plan_load_lane": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vsd_plan_info": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "vsd_plan_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_plan_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vsd_plan_infer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsd_plan_load_prompt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p]),
    "vsd_plan_free": (None, [C.c_void_p, C.c_void_p]),
    "vsd_pinned_alloc": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "vsd_pinned_free": (None, [C.c_void_p, C.c_void_p]),
    "vsd_profile_begin": (C.c_int, [C.c_void_p]),
    "vsd_profile_end": (C.c_int, [C.c_void_p]),
    "vsd_profile_overhead": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "vsd_stage_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """dlopen libvsd.so and bind every symbol.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m videosd_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # torch first: libvsd.so and torch must share ONE HIP runtime in the process. torch ships its own libamdhip64; if
    # libvsd.so is opened before it, the loader binds /opt/rocm's copy instead, torch then brings a second runtime and
    # vsd_create() sees no device (the streams / device pointers torch hands over belong to the other runtime).
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    rebuild = f"{LIB_PATH} is stale: rebuild it with `python -m videosd_amd.build --force`"
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # a declared symbol that is not exported = a library built from older sources
        except AttributeError:
            raise RuntimeError(f"{rebuild} (symbol {name} is missing)") from None
        fn.restype = res
        fn.argtypes = args
    # same interface version and the same vsd_conv_desc layout on both sides of the boundary (the struct grew
    # three fields while vsd_version stayed 1; a stale library was only caught through missing symbols)
    if lib.vsd_version() != VERSION or lib.vsd_conv_desc_size() != C.sizeof(ConvDesc):
        raise RuntimeError(f"{rebuild} (library: interface version {lib.vsd_version()}, vsd_conv_desc {lib.vsd_conv_desc_size()} bytes; "
                           f"this binding: version {VERSION}, {C.sizeof(ConvDesc)} bytes)")
    _lib = lib
    return lib


class Context:
    """One vsd_ctx (one GPU).  Methods raise RuntimeError with vsd_last_error() on failure."""

    def __init__(self, device_id: int = 0):
        self.lib = load()
        self.h = self.lib.vsd_create(device_id)
        if not self.h:
            raise RuntimeError(f"vsd_create({device_id}) failed: no such HIP device")

    def close(self):
        if getattr(self, "h", None):
            self.lib.vsd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.vsd_last_error(self.h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def call(self, name: str, *args):
        self.check(getattr(self.lib, name)(self.h, *args), name)

    def stage_times(self):
        ms = (C.c_float * len(FAMILIES))()
        n = (C.c_int64 * len(FAMILIES))()
        fl = (C.c_double * len(FAMILIES))()
        self.call("vsd_stage_times", ms, n, fl)
        return {f: {"ms": float(ms[i]), "launches": int(n[i]), "flops": float(fl[i])} for i, f in enumerate(FAMILIES)}
