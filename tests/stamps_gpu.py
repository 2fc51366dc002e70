"""Diagnostic (not a test): phase time stamps of workgroup 0 of one kernel launch of the ELBO step.

usage: python tests/stamps_gpu.py [workload] tag [tag ...]     tag = kind:group[:workgroup], kinds fwd dx dw pdx
Phases: fwd  0 top 1 vm-wait 2 B1 3 derived 4 B2 5 mfma 6 reduce 7 epilogue
        dx   0 top 1 vm-wait 2 B1 3 mask 4 B2 5 mfma|issue 6 epilogue
        dw   0 top(loads issued at 1) 1 loads issued 2 barrier 3 staged 4 barrier 5 pooled 6 mfma
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from bayesrul_amd.engine import AdamHyper, SviEngine  # noqa: E402

KIND = {"fwd": 0, "dx": 1, "dw": 2, "pdx": 7}
args = sys.argv[1:]
wl_name = "flipout_conv_s10"
if args and args[0] in bench.WORKLOADS:
    wl_name = args.pop(0)
wl = bench.WORKLOADS[wl_name]
S, B = wl["S"], wl["B"]
eng = SviEngine(net=wl["net"], guide=wl["guide"], fit_context=wl["fit_context"], prec="bf16x3", max_particles=S,
                max_batch=B)
eng.init_params(bench.mu0_for(wl["net"]), wl["q_scale"])
x, y = bench.synth(B)
x, y = x.cuda(), y.cuda()
hyp = AdamHyper(lr=wl["lr"])
lib = eng.lib
lib.bnn_debug_stamps.argtypes = [C.c_void_p, C.c_int]
lib.bnn_debug_stamps_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for _ in range(3):
    eng.step(x, y, S, bench.N_DATA, 0.0, wl["prior_scale"], hyp, seed=1)
torch.cuda.synchronize()
NW = 16
for tag in args:
    parts = tag.split(":")
    kind, grp = parts[0], parts[1]
    lib.bnn_debug_stamps_block.argtypes = [C.c_void_p, C.c_int]
    assert lib.bnn_debug_stamps_block(eng._plan, int(parts[2]) if len(parts) > 2 else 0) == 0
    assert lib.bnn_debug_stamps(eng._plan, KIND[kind] * 16 + int(grp)) == 0
    eng.step(x, y, S, bench.N_DATA, 0.0, wl["prior_scale"], hyp, seed=1)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * (NW * 48 * 8))()
    assert lib.bnn_debug_stamps_read(eng._plan, buf, C.sizeof(buf)) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, 48, 8).astype(np.int64)
    print(f"==== {wl_name} {tag}")
    live = [w for w in range(NW) if a[w, 0, 0] > 0]
    if not live:
        print("  no stamps")
        continue
    t0 = min(a[w, 0, 0] for w in live)
    nk = int(max((a[w, :, 0] > 0).sum() for w in live))
    print("  waves", live, "iterations stamped", nk)
    for w in live:
        k1, k2 = 2, min(nk - 2, 12)
        if k2 <= k1:
            k1, k2 = 0, nk - 1
        per = (a[w, k2, 0] - a[w, k1, 0]) / max(1, k2 - k1)
        # mean offset of each phase from the iteration top over iterations k1..k2
        offs = []
        for ph in range(1, 8):
            v = a[w, k1:k2, ph]
            m = v > 0
            offs.append(int(((v - a[w, k1:k2, 0])[m]).mean()) if m.any() else -1)
        print(f"  wave {w:2d}: first top {a[w, 0, 0] - t0:7d}  period {per:8.1f}  phase offsets {offs}")
    lib.bnn_debug_stamps(eng._plan, -1)
