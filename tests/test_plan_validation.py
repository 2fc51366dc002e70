"""Host-side plan validation (no GPU): every launch geometry the library would use for (net, estimator, precision, S, B)
is built and checked by `bnn_plan_validate` — DMA instruction counts against LDS plane / slot sizes, slot rings against
windows in flight, counted-wait ranges, LDS budgets.  Replaces round 1's rerun-based race screen."""
import ctypes as C
import itertools

import pytest

from bayesrul_amd import _native as N


@pytest.fixture(scope="module")
def lib():
    from bayesrul_amd.csrc.build import build
    build()
    return N.load()


GEOMS = [(1, 1), (1, 33), (1, 100), (2, 7), (2, 257), (10, 1000), (20, 1000), (3, 33)]
# the fp32 dense forward / dX deal (particle, chunk, row step) items to <= 256 workgroups (equal ranges or full ranges + packed
# remainders): validation enumerates every workgroup's segments on the host and refuses a schedule that drops or repeats an item
DENSE_GEOMS = [(1, 1000), (2, 100), (3, 500), (5, 250), (7, 333), (10, 100), (10, 250), (10, 500), (12, 1000), (13, 999),
               (16, 640), (25, 1000), (26, 64), (40, 1000), (64, 32), (100, 10)]


@pytest.mark.parametrize("net,prec", [("inception", N.PREC_BF16X3), ("inception", N.PREC_F32), ("linear", N.PREC_BF16X3),
                                      ("linear", N.PREC_F32)])
def test_every_launch_geometry_validates(lib, net, prec):
    for mode, (S, B) in itertools.product((N.MODE_NORMAL, N.MODE_LRT, N.MODE_FLIPOUT, N.MODE_RADIAL), GEOMS):
        d = N.PlanDesc(0 if net == "inception" else 1, mode, prec, S, B, 30, 18, 0)
        p = C.c_void_p()
        if net == "inception" and prec == N.PREC_BF16X3 and mode == N.MODE_LRT:
            # LRT on the Inception net exists on the exact-fp32 plan only: refused with a message at plan creation
            assert lib.bnn_plan_create(C.byref(d), C.byref(p)) == -1 and b"exact-fp32" in lib.bnn_last_error()
            continue
        N.check(lib.bnn_plan_create(C.byref(d), C.byref(p)))
        try:
            for train in (1, 0):
                rc = lib.bnn_plan_validate(p, -1, S, B, train)
                assert rc == 0, (net, prec, mode, S, B, train, lib.bnn_last_error())
            # validation / predict run the plain estimator on the same plan
            plain = N.MODE_RADIAL if mode == N.MODE_RADIAL else N.MODE_NORMAL
            assert lib.bnn_plan_validate(p, plain, S, B, 0) == 0, lib.bnn_last_error()
        finally:
            lib.bnn_plan_destroy(p)


def test_predictive_pass_geometry_and_refusals(lib):
    # configs[4]: 10 particles x 10,000 windows per chunk
    d = N.PlanDesc(0, N.MODE_FLIPOUT, N.PREC_BF16X3, 100, 10000, 30, 18, 100000)
    p = C.c_void_p()
    N.check(lib.bnn_plan_create(C.byref(d), C.byref(p)))
    assert lib.bnn_plan_validate(p, N.MODE_NORMAL, 10, 10000, 0) == 0, lib.bnn_last_error()
    # more windows than the plan holds: refused with a message, not launched
    assert lib.bnn_plan_validate(p, N.MODE_NORMAL, 11, 10000, 0) == -1 and b"capacity" in lib.bnn_last_error()
    lib.bnn_plan_destroy(p)
    # windows longer than the 32-row tile are refused at plan creation
    d = N.PlanDesc(0, N.MODE_FLIPOUT, N.PREC_BF16X3, 1, 4, 31, 18, 0)
    assert lib.bnn_plan_create(C.byref(d), C.byref(p)) == -1
    # a Flipout plan overridden to LRT per call: refused at the call on the split-bf16 plan
    d = N.PlanDesc(0, N.MODE_FLIPOUT, N.PREC_BF16X3, 2, 8, 30, 18, 0)
    N.check(lib.bnn_plan_create(C.byref(d), C.byref(p)))
    assert lib.bnn_plan_validate(p, N.MODE_LRT, 2, 8, 1) == -1 and b"exact-fp32" in lib.bnn_last_error()
    lib.bnn_plan_destroy(p)
    # the fused trunk kernels address rows with 32-bit byte offsets
    d = N.PlanDesc(0, N.MODE_FLIPOUT, N.PREC_BF16X3, 100, 10000, 30, 18, 0)
    N.check(lib.bnn_plan_create(C.byref(d), C.byref(p)))
    assert lib.bnn_plan_validate(p, -1, 100, 10000, 0) == -1 and b"32-bit" in lib.bnn_last_error()
    lib.bnn_plan_destroy(p)


@pytest.mark.parametrize("prec", [N.PREC_BF16X3, N.PREC_F32])
def test_more_than_256_particles_keep_their_dw_slabs_apart(lib, prec):
    """The trunk dW kernels write one partial image per workgroup: S * nsplit of them, nsplit = max(1, min(B, 256 / S)).
    Round 2 sized the slab regions for a fixed 256 / 256 / 512: S = 300 ran into the next region, S = 600 past the end
    (ADVICE r02).  They are sized from max_particles now; a call that would still exceed them is refused."""
    for S, B in ((300, 4), (600, 2)):
        d = N.PlanDesc(0, N.MODE_FLIPOUT, prec, S, B, 30, 18, 0)
        p = C.c_void_p()
        N.check(lib.bnn_plan_create(C.byref(d), C.byref(p)))
        try:
            assert lib.bnn_plan_validate(p, -1, S, B, 1) == 0, lib.bnn_last_error()
            ws = C.c_size_t()
            N.check(lib.bnn_plan_workspace_bytes(p, C.byref(ws)))
            assert ws.value > 0
        finally:
            lib.bnn_plan_destroy(p)


def test_fp32_dense_item_schedules_cover_every_item_once(lib):
    for S, B in GEOMS + DENSE_GEOMS:
        for mode in (N.MODE_FLIPOUT, N.MODE_LRT, N.MODE_RADIAL):
            d = N.PlanDesc(0, mode, N.PREC_F32, S, B, 30, 18, 0)
            p = C.c_void_p()
            N.check(lib.bnn_plan_create(C.byref(d), C.byref(p)))
            try:
                for train in (1, 0):
                    assert lib.bnn_plan_validate(p, -1, S, B, train) == 0, (S, B, mode, train, lib.bnn_last_error())
            finally:
                lib.bnn_plan_destroy(p)
