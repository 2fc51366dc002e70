"""Diagnostic (not a test): per-site gradient error of the bf16x3 Inception step against the oracle and
run-to-run reproducibility (two fresh engines, identical inputs and injected noise), all estimators."""
import sys

import torch

sys.path.insert(0, ".")
from oracle import restatement as R  # noqa: E402,F401
from tests.noise_util import rel_l2, to_injected  # noqa: E402
from tests.test_gpu_parity import N_DATA, _setup  # noqa: E402

for mode in ("flipout", "lrt", "radial"):
    S, B = 2, 100
    runs = []
    for rep in range(2):
        eng, cfg, st_, x, y, noise, (ps, qs, lr) = _setup("inception", mode, "bf16x3", S, B, q_boost=5.0)
        if rep == 0:
            st = st_
            loss_o, aux = st.loss_and_grads(x, y, noise)
        inj = to_injected(eng, cfg, noise, B)
        res, preds = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj, want_preds=True)
        torch.cuda.synchronize()
        runs.append((float(res[0]), eng.grad.cpu().clone(), preds.cpu().clone()))
    (l0, g0, p0), (l1, g1, p1) = runs
    print(f"== {mode}: loss {l0:.9g} / {l1:.9g}  oracle {float(loss_o):.9g}  rel err {abs(l0 - float(loss_o)) / abs(float(loss_o)):.2e}")
    print(f"   run-to-run: loss diff {abs(l0 - l1):.3e}  preds max diff {float((p0 - p1).abs().max()):.3e}  "
          f"grad rel-L2 {rel_l2(g1, g0.double()):.3e}")
    worst = 0.0
    for s, off, num in eng.sites:
        e = rel_l2(g0[off:off + num], st.mu[s].grad)
        r = rel_l2(g0[eng.P + off:eng.P + off + num], st.rho[s].grad)
        worst = max(worst, e)
        print(f"   {s:32s} mu {e:.2e}  rho {r:.2e}")
    print(f"   worst mu error {worst:.3e}")
