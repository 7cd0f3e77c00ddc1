"""The plan file's fixed points that need no GPU: the generated dispatch of csrc/plan.hip is the one scripts/gen_plan_dispatch.py
makes from lib.py's signatures today, and every entry point a plan may hold is declared."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_the_committed_dispatch_is_what_the_generator_writes_today():
    import gen_plan_dispatch

    committed = open(os.path.join(ROOT, "videosd_amd", "csrc", "plan_dispatch.inc")).read()
    assert committed == gen_plan_dispatch.generate(), "run python scripts/gen_plan_dispatch.py"


def test_every_entry_point_of_a_plan_is_declared_and_takes_the_context_first():
    import ctypes as C

    from videosd_amd import lib as L
    from videosd_amd.plan import PLAN_FUNCS, _PTR_FIELDS

    header = open(os.path.join(ROOT, "include", "vsd.h")).read()
    for name in PLAN_FUNCS:
        assert name in L.SIGNATURES and f"{name}(" in header, name
        assert L.SIGNATURES[name][1][0] is C.c_void_p
    assert len(PLAN_FUNCS) == len(set(PLAN_FUNCS))
    # every pointer field of the conv descriptor is known to the exporter (a new one must be patched at load, too)
    assert {n for n, _ in _PTR_FIELDS} == {n for n, t in L.ConvDesc._fields_ if t is C.c_void_p}
    for f in ("vsd_plan_load", "vsd_plan_info", "vsd_plan_infer", "vsd_plan_free"):
        assert f in L.SIGNATURES and f"{f}(" in header
