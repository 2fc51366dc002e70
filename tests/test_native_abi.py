"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/*.h declares,
the ctypes structs match the compiled structs, and plan tables agree with the reference's
state_dict layout.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from bayesrul_amd import _native as N
from oracle import restatement as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from bayesrul_amd.csrc.build import build
    build()
    return N.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "bayesrul_amd.h")).read()
    decl = set(re.findall(r"\b(bnn_[a-z_0-9]+)\s*\(", hdr))
    assert decl, "no declarations parsed"
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert decl == set(N.EXPORTS), decl ^ set(N.EXPORTS)


def test_struct_sizes_match(lib):
    for i, st in enumerate(N._ABI_STRUCTS):
        assert lib.bnn_abi_sizeof(i) == C.sizeof(st)
    assert lib.bnn_version() == N.ABI_VERSION == 3


@pytest.mark.parametrize("net", ["inception", "linear"])
def test_plan_tables_match_reference_layout(lib, net):
    d = N.PlanDesc(N.NETS[net] if hasattr(N, "NETS") else (0 if net == "inception" else 1), N.MODE_LRT, N.PREC_F32,
                   2, 16, 30, 18, 0)
    p = C.c_void_p()
    N.check(lib.bnn_plan_create(C.byref(d), C.byref(p)))
    P = C.c_int64()
    N.check(lib.bnn_plan_num_params(p, C.byref(P)))
    assert P.value == R.n_params(net)
    n = C.c_int32()
    N.check(lib.bnn_plan_num_sites(p, C.byref(n)))
    shapes = R.site_shapes(net)
    assert n.value == len(shapes)
    off = 0
    for i, (name, shp) in enumerate(shapes):
        nm, o, num = C.c_char_p(), C.c_int64(), C.c_int64()
        N.check(lib.bnn_plan_site(p, i, C.byref(nm), C.byref(o), C.byref(num)))
        numel = 1
        for s in shp:
            numel *= s
        assert (nm.value.decode(), o.value, num.value) == (name, off, numel)
        off += numel
    lib.bnn_plan_destroy(p)


def test_errors_are_reported_not_thrown(lib):
    d = N.PlanDesc(7, 0, 0, 1, 1, 30, 18, 0)
    p = C.c_void_p()
    rc = lib.bnn_plan_create(C.byref(d), C.byref(p))
    assert rc == -1 and b"unknown net" in lib.bnn_last_error()
    with pytest.raises(N.NativeError):
        N.check(rc)
    # unbound plan refuses compute
    d = N.PlanDesc(0, 1, 0, 1, 4, 30, 18, 0)
    N.check(lib.bnn_plan_create(C.byref(d), C.byref(p)))
    a = N.ElboArgs()
    a.batch, a.particles = 4, 1
    rc = lib.bnn_elbo_evaluate(p, C.byref(a), None, None, None)
    assert rc == -2
    lib.bnn_plan_destroy(p)


def test_engine_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayesrul_amd.engine import SviEngine
    with pytest.raises(RuntimeError):
        SviEngine(net="inception")
