"""Diagnostic (not a test): per-site gradient error of the bf16x3 Inception step against the oracle,
for the fused conv dX kernel (default) and the previous one (BNN_DX_V1=1), all estimators."""
import os
import sys

import torch

sys.path.insert(0, ".")
from oracle import restatement as R  # noqa: E402
from tests.noise_util import rel_l2, to_injected  # noqa: E402
from tests.test_gpu_parity import N_DATA, _setup  # noqa: E402

for mode in ("flipout", "lrt", "radial"):
    S, B = 2, 100
    grads = {}
    for v1 in (True, False):
        if v1:
            os.environ["BNN_DX_V1"] = "1"
        else:
            os.environ.pop("BNN_DX_V1", None)
        eng, cfg, st_, x, y, noise, (ps, qs, lr) = _setup("inception", mode, "bf16x3", S, B, q_boost=5.0)
        if v1:
            st = st_
        inj = to_injected(eng, cfg, noise, B)
        res, preds = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj, want_preds=True)
        torch.cuda.synchronize()
        grads[v1] = eng.grad.cpu().clone()
        if v1:
            loss_o, aux = st.loss_and_grads(x, y, noise)
        print(f"{mode} v1={v1} loss {float(res[0]):.8g} oracle {float(loss_o):.8g}")
    print(f"== {mode}: per-site rel-L2 vs oracle   [mu old, mu new | rho old, rho new]   new-vs-old mu")
    worst = 0.0
    for s, off, num in eng.sites:
        gm, gr = st.mu[s].grad, st.rho[s].grad
        e = [rel_l2(grads[v][off:off + num], gm) for v in (True, False)]
        r = [rel_l2(grads[v][eng.P + off:eng.P + off + num], gr) for v in (True, False)]
        d = rel_l2(grads[False][off:off + num], grads[True][off:off + num].double())
        worst = max(worst, e[1])
        print(f"  {s:32s} {e[0]:.2e} {e[1]:.2e} | {r[0]:.2e} {r[1]:.2e}   {d:.2e}")
    print(f"  worst new mu error {worst:.3e}")
