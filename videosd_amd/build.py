"""Build libvsd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvsd.so")
# the implicit-GEMM conv kernel is a template with ~60 instantiations: one translation unit per tile family, so that
# the families compile in parallel (as one file the library took 4.5 minutes to build)
SOURCES = ["api.hip", "plan.hip", "conv_gemm.hip", "conv_t128x128.hip", "conv_t128x64.hip", "conv_t64x64.hip", "conv_t64x128.hip",
           "conv_t256x128.hip", "conv_t256x256.hip", "conv_halo.hip", "conv_c64.hip", "fused_tail.hip", "norm.hip", "attention.hip", "elementwise.hip", "prompt_fold.hip"]
HEADERS = ["common.h", "conv_kernels.h", "conv_epilogue.inc", "conv_gemm_body.inc", "plan_dispatch.inc", os.path.join("..", "..", "include", "vsd.h")]
# attention keeps its O / S accumulators live across the key loop and touches them with VALU every tile (online-softmax
# rescale, exp): with the default AGPR placement the compiler moves them through v_accvgpr_read/write every tile
# (~190 of 1300 instructions in the d=40 kernel); VGPR-form MFMA operands remove those moves.
# conv_t256x128: 128 accumulator registers + 96 of fragments per wave; in the default (AGPR) form the compiler shuffled the
# accumulators through AGPR copies inside the interleaved main loop (340 v_accvgpr moves per K tile, 396 registers);
# VGPR form: 230 / 274 registers, 5 moves.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "conv_t256x128.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "conv_t256x256.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "conv_c64.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _obj(f: str) -> str:
    return os.path.join(HERE, "build", f.replace(".hip", ".o"))


def _stale(f: str) -> bool:
    o = _obj(f)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    deps = [os.path.join(CSRC, f)] + [os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return not os.path.exists(LIB) or any(_stale(f) or os.path.getmtime(_obj(f)) > os.path.getmtime(LIB) for f in SOURCES)


def build(force: bool = False, verbose: bool = True, jobs: int = 0) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    todo = [f for f in SOURCES if force or _stale(f)]
    jobs = jobs or min(len(todo), os.cpu_count() or 4) or 1
    running = []

    def reap(block_until: int):
        while len(running) > block_until:
            p, f = running.pop(0)
            if p.wait() != 0:
                for q, _ in running:
                    q.kill()
                raise RuntimeError(f"hipcc failed on {f}")

    for f in todo:
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"] + EXTRA_FLAGS.get(f, []) + ["-c", os.path.join(CSRC, f), "-o", _obj(f)]
        if verbose:
            print(" ".join(cmd), flush=True)
        running.append((subprocess.Popen(cmd), f))
        reap(jobs - 1)
    reap(0)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_probe(verbose: bool = False) -> str:
    """scripts/conv_probe.bin: the conv kernels built with -DVSD_CONV_PROBE (in-kernel cycle stamps) + the probe driver."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    root = os.path.dirname(HERE)
    out = os.path.join(root, "scripts", "conv_probe.bin")
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    srcs = [f for f in SOURCES if f.startswith("conv_") or f == "api.hip"]
    procs, objs = [], []
    for f in srcs:
        o = os.path.join(HERE, "build", "probe_" + f.replace(".hip", ".o"))
        objs.append(o)
        procs.append((subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-DVSD_CONV_PROBE", "-w", "-c",
                                        os.path.join(CSRC, f), "-o", o]), f))
    for p, f in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {f}")
    drv = os.path.join(HERE, "build", "probe_main.o")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-DVSD_CONV_PROBE", "-w", "-c",
                           os.path.join(root, "scripts", "conv_probe.cpp"), "-o", drv])
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-o", out, drv] + objs)
    return out


def build_timeline(verbose: bool = False) -> str:
    """videosd_amd/libvsd_tl.so: the whole library built with -DVSD_WG_TIMELINE (csrc/common.h): the conv kernels log every
    workgroup's start / end time and placement.  Development tool (scripts/wg_timeline.py loads it through VSD_LIB)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(HERE, "libvsd_tl.so")
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs, objs = [], []
    for f in SOURCES:
        o = os.path.join(HERE, "build", "tl_" + f.replace(".hip", ".o"))
        objs.append(o)
        deps = [os.path.join(CSRC, f)] + [os.path.join(CSRC, h) for h in HEADERS]
        if os.path.exists(o) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps):
            continue  # (up to date)
        procs.append((subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-DVSD_WG_TIMELINE", "-w"] +
                                       EXTRA_FLAGS.get(f, []) + ["-c", os.path.join(CSRC, f), "-o", o]), f))
    for p, f in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {f}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_variant(tag: str, defines, verbose: bool = False) -> str:
    """videosd_amd/libvsd_<tag>.so: the library with extra -D defines (experiment builds: A/B through VSD_LIB)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(HERE, f"libvsd_{tag}.so")
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs, objs = [], []
    for f in SOURCES:
        o = os.path.join(HERE, "build", f"{tag}_" + f.replace(".hip", ".o"))
        objs.append(o)
        procs.append((subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-w"] + [f"-D{d}" for d in defines] + EXTRA_FLAGS.get(f, []) +
                                       ["-c", os.path.join(CSRC, f), "-o", o]), f))
    for p, f in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {f}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_whatif(verbose: bool = False) -> str:
    """videosd_amd/libvsd_probe.so: the library built with -DVSD_PROBE -- environment switches that REMOVE a cost (VSD_SKIP_GN /
    VSD_SKIP_REDUCE / VSD_SKIP_ATTN: the kernels are not launched; VSD_SKIP_EPI: the unsplit GEMM-form layers leave before their
    epilogue).  Results are garbage: scripts/whatif_probe.sh reads the frame rate only (VSD_LIB selects the library)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(HERE, "libvsd_probe.so")
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs, objs = [], []
    for f in SOURCES:
        o = os.path.join(HERE, "build", "whatif_" + f.replace(".hip", ".o"))
        objs.append(o)
        procs.append((subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-DVSD_PROBE", "-w"] + EXTRA_FLAGS.get(f, []) +
                                       ["-c", os.path.join(CSRC, f), "-o", o]), f))
    for p, f in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {f}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m videosd_amd.build --variant nt VSD_NT_STORES
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    elif "--whatif" in sys.argv:
        print(build_whatif())
    elif "--timeline" in sys.argv:
        print(build_timeline())
    elif "--probe" in sys.argv:
        print(build_probe())
    else:
        build(force="--force" in sys.argv)
