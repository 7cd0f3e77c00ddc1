"""Build libvsd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvsd.so")
SOURCES = ["api.hip", "conv_gemm.hip", "norm.hip", "attention.hip", "elementwise.hip"]
# attention keeps its O / S accumulators live across the key loop and touches them with VALU every tile (online-softmax
# rescale, exp): with the default AGPR placement the compiler moves them through v_accvgpr_read/write every tile
# (~190 of 1300 instructions in the d=40 kernel); VGPR-form MFMA operands remove those moves.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "vsd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for f in SOURCES:
        o = os.path.join(HERE, "build", f.replace(".hip", ".o"))
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"] + EXTRA_FLAGS.get(f, []) + ["-c", os.path.join(CSRC, f), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), f))
        objs.append(o)
    for p, f in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {f}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
