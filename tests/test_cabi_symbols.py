"""The C-ABI library loads (no GPU needed) and exports every function include/vsd.h declares; the ctypes
struct mirrors the C struct field for field."""
import ctypes
import os
import re

from videosd_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "vsd.h")).read()


def test_every_declared_symbol_is_exported_and_bound():
    if not os.path.exists(L.LIB_PATH):
        from videosd_amd import build

        build.build(verbose=False)
    lib = L.load()
    declared = set(re.findall(r"\b(vsd_[a-z0-9_]+)\s*\(", _header()))
    assert len(declared) >= 20
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    # the library and the binding agree on the interface version and on the struct layout (lib.load() refuses a stale build)
    assert lib.vsd_version() == L.VERSION == int(re.search(r"#define VSD_VERSION (\d+)", _header()).group(1))
    assert lib.vsd_conv_desc_size() == ctypes.sizeof(L.ConvDesc)
    assert lib.vsd_create(10_000) is None  # no such device -> NULL, never aborts


def test_conv_desc_struct_matches_header_field_order():
    body = re.search(r"typedef struct vsd_conv_desc \{(.*?)\} vsd_conv_desc;", _header(), re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split()
        typ_is_ptr = "*" in decl
        for n in decl.split(",") if "," in decl else [decl]:
            nm = n.replace("*", " ").split()[-1]
            fields.append((nm, typ_is_ptr))
    got = [(n, issubclass(t, ctypes.c_void_p) or t is ctypes.c_void_p) for n, t in L.ConvDesc._fields_]
    assert [f[0] for f in fields] == [g[0] for g in got]
    assert fields == got


def test_one_hip_runtime_in_the_process_whatever_the_import_order():
    """libvsd.so opened before torch used to bind /opt/rocm's libamdhip64 while torch brought its own: two runtimes,
    vsd_create() found no device (build() followed by smoke() in one process). lib.load() imports torch first."""
    import re
    import subprocess
    import sys

    code = ("from videosd_amd import lib; lib.load(); import torch, re; "
            "print(sorted(set(re.findall(r'/\\S*libamdhip64\\S*', open('/proc/self/maps').read()))))")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         cwd=__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    libs = re.findall(r"'([^']+)'", out.stdout.strip().splitlines()[-1])
    assert len(libs) == 1, libs


def test_a_stale_library_is_refused_with_the_rebuild_command(monkeypatch):
    """vsd_conv_desc grew while vsd_version stayed the same; a stale libvsd.so must be named as such at load."""
    import pytest

    L.load()
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "VERSION", L.VERSION + 1)
    with pytest.raises(RuntimeError, match="videosd_amd.build --force"):
        L.load()
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "VERSION", L.VERSION - 1)
    assert L.load() is not None
