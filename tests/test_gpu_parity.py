"""GPU parity: the HIP path (through the C ABI) against the oracle on identical injected noise.

f32 plans use the exact-fp32 MFMA (tight tolerances); bf16x3 plans use split-bf16 on the
forward mean path (ELBO tolerance 1e-3 relative is the north star's; we hold 1e-4) and single
bf16 in the backward contractions (gradient tolerance is a relative L2 error).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import restatement as R
from tests.noise_util import record, flat, from_injected, oracle_cfg, rel_l2, synth_batch, to_injected

HYP = {"lrt": (0.138793, 0.001351, 8.57e-4), "flipout": (0.198768, 0.000214, 9.48e-4),
       "radial": (0.092516, 0.001241, 9.56e-4), "normal": (0.15, 0.002, 9e-4)}
N_DATA = 238200


def _engine(net, mode, prec, S, B, **kw):
    from bayesrul_amd.engine import SviEngine
    guide = "radial" if mode == "radial" else "normal"
    ctx = mode if mode in ("lrt", "flipout") else None
    return SviEngine(net=net, guide=guide, fit_context=ctx, prec=prec, max_particles=S, max_batch=B, **kw)


def _setup(net, mode, prec, S, B, q_boost=1.0, seed=0):
    ps, qs, lr = HYP[mode]
    qs *= q_boost
    eng = _engine(net, mode, prec, S, B)
    mu0 = R.init_mu0(net, seed, torch.float64)
    eng.init_params(mu0, qs)
    cfg = oracle_cfg(net, mode, ps)
    st = R.SviState(cfg, mu0, qs, R.AdamConfig(lr=lr))
    x, y = synth_batch(B)
    noise = R.make_noise(cfg, B, S, torch.Generator().manual_seed(4321))
    return eng, cfg, st, x, y, noise, (ps, qs, lr)


# bf16x3 plan, gradients against the f64 oracle (rel-L2): bounds = 2 x the values measured on MI355X (gpurun_out/
# measured_errors.jsonl of the round-3 run, printed by tests.noise_util.record), per estimator: (d mu, d rho).  The backward
# contractions are single bf16; the exact-fp32 plan - the default and the judged one - is held to 1e-3 per site.
# measured (S = 2, B = 100, shipped hyper-parameters): whole-vector d mu 1.3e-3 .. 1.6e-3, d rho 2.4e-4 (radial), 3.0e-3 (flipout);
# worst single site (q_scale x 5) d mu 4.1e-3 .. 8.6e-3, d rho 5.6e-3 (radial) .. 2.2e-2 (flipout, layers.1.branch2.0.weight)
GRAD_TOL = {"flipout": (3.1e-3, 6.1e-3), "radial": (2.6e-3, 5e-4)}
SITE_TOL = {"flipout": (1.4e-2, 4.5e-2), "radial": (8.2e-3, 1.2e-2)}


@pytest.mark.parametrize("mode", ["lrt", "flipout", "radial", "normal"])
@pytest.mark.parametrize("net", ["inception", "linear"])
def test_step_f32_matches_oracle(net, mode):
    """loss / kl / loglik / preds / d(mu, rho) of one svi.step, exact-fp32 MFMA path.
    q_scale x20 so that every variance / perturbation term is far above fp32 round-off."""
    S, B = 2, 7
    eng, cfg, st, x, y, noise, (ps, qs, lr) = _setup(net, mode, "f32", S, B, q_boost=20.0)
    inj = to_injected(eng, cfg, noise, B)
    res, preds = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj, want_preds=True)
    loss_o, aux = st.loss_and_grads(x, y, noise)
    res = res.cpu().double()
    assert abs(float(res[0]) - float(loss_o)) <= 2e-5 * abs(float(loss_o)), (float(res[0]), float(loss_o))
    assert abs(float(res[1]) - float(aux["kl"])) <= 2e-5 * abs(float(aux["kl"]))
    assert abs(float(res[2]) - float(aux["loglik"])) <= 2e-5 * abs(float(aux["loglik"]))
    assert torch.allclose(preds.cpu().double(), aux["preds"], rtol=2e-4, atol=1e-5)
    g = eng.grad.cpu()
    gmu = torch.cat([st.mu[s].grad.flatten() for s, _ in R.site_shapes(net)])
    grho = torch.cat([st.rho[s].grad.flatten() for s, _ in R.site_shapes(net)])
    assert rel_l2(g[:eng.P], gmu) < 2e-4, rel_l2(g[:eng.P], gmu)
    assert rel_l2(g[eng.P:2 * eng.P], grho) < 2e-4, rel_l2(g[eng.P:2 * eng.P], grho)
    # per-site check so that a small site cannot hide behind a big one
    for s, off, num in eng.sites:
        assert rel_l2(g[off:off + num], st.mu[s].grad) < 1e-3, ("mu", s)
        assert rel_l2(g[eng.P + off:eng.P + off + num], st.rho[s].grad) < 1e-3, ("rho", s)


@pytest.mark.parametrize("mode", ["flipout", "radial"])
def test_step_bf16x3_elbo_within_tolerance(mode):
    """north star: ELBO within 1e-3 relative of the reference arithmetic (we hold 1e-5) at the
    shipped hyper-parameters, batch 100."""
    S, B = 2, 100
    eng, cfg, st, x, y, noise, (ps, qs, lr) = _setup("inception", mode, "bf16x3", S, B)
    inj = to_injected(eng, cfg, noise, B)
    res, preds = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj, want_preds=True)
    loss_o, aux = st.loss_and_grads(x, y, noise)
    res = res.cpu().double()
    # measured 1.1e-6 .. 3.3e-6 (north star: 1e-3)
    assert abs(float(res[0]) - float(loss_o)) <= 1e-5 * abs(float(loss_o)), (float(res[0]), float(loss_o))
    assert torch.allclose(preds.cpu().double(), aux["preds"], rtol=2e-3, atol=1e-4)
    g = eng.grad.cpu()
    gmu = torch.cat([st.mu[s].grad.flatten() for s, _ in R.site_shapes("inception")])
    grho = torch.cat([st.rho[s].grad.flatten() for s, _ in R.site_shapes("inception")])
    record(f"step_bf16x3_elbo[{mode}]", loss_rel=abs(float(res[0]) - float(loss_o)) / abs(float(loss_o)),
           dmu=rel_l2(g[:eng.P], gmu), drho=rel_l2(g[eng.P:2 * eng.P], grho))
    assert rel_l2(g[:eng.P], gmu) < GRAD_TOL[mode][0], rel_l2(g[:eng.P], gmu)
    assert rel_l2(g[eng.P:2 * eng.P], grho) < GRAD_TOL[mode][1], rel_l2(g[eng.P:2 * eng.P], grho)


@pytest.mark.parametrize("mode", ["flipout", "radial"])
def test_step_bf16x3_per_site_gradients_and_reproducibility(mode):
    """bf16x3 plan: every site's gradient (a small site must not hide behind the dense layer's 82 % of the
    weights; catches a wrong conv dX / dW / pooled scatter) and run-to-run reproducibility: two fresh engines on
    identical inputs and injected noise give the same loss and predictions bit for bit (a race in an LDS-DMA
    pipeline shows up here) and gradients equal up to the order of the fp32 atomics."""
    S, B = 2, 100
    runs = []
    for rep in range(2):
        eng, cfg, st_, x, y, noise, (ps, qs, lr) = _setup("inception", mode, "bf16x3", S, B, q_boost=5.0)
        if rep == 0:
            st = st_
            st.loss_and_grads(x, y, noise)
        inj = to_injected(eng, cfg, noise, B)
        res, preds = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj, want_preds=True)
        runs.append((float(res[0]), eng.grad.cpu().clone(), preds.cpu().clone()))
    (l0, g0, p0), (l1, g1, p1) = runs
    assert l0 == l1 and torch.equal(p0, p1)
    assert rel_l2(g1, g0.double()) < 1e-6
    emu = {s: rel_l2(g0[off:off + num], st.mu[s].grad) for s, off, num in eng.sites}
    erho = {s: rel_l2(g0[eng.P + off:eng.P + off + num], st.rho[s].grad) for s, off, num in eng.sites}
    record(f"step_bf16x3_per_site[{mode}]", worst_mu=max(emu.values()), worst_mu_site=max(emu, key=emu.get),
           worst_rho=max(erho.values()), worst_rho_site=max(erho, key=erho.get), run_to_run=rel_l2(g1, g0.double()))
    for s, off, num in eng.sites:
        assert emu[s] < SITE_TOL[mode][0], ("mu", s, emu[s])
        assert erho[s] < SITE_TOL[mode][1], ("rho", s, erho[s])


@pytest.mark.parametrize("mode", ["lrt", "radial"])
def test_three_adam_steps_match_oracle(mode):
    """svi.step x3 incl. ClippedAdam on (mu, log sigma): parameters after 3 steps."""
    S, B = 1, 9
    eng, cfg, st, x, y, _, (ps, qs, lr) = _setup("inception", mode, "f32", S, B, q_boost=20.0)
    from bayesrul_amd.engine import AdamHyper
    hyp = AdamHyper(lr=lr, betas=(0.95, 0.999), clip_norm=15.0)
    for k in range(3):
        noise = R.make_noise(cfg, B, S, torch.Generator().manual_seed(100 + k))
        inj = to_injected(eng, cfg, noise, B)
        res = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, hyp, noise=inj)
        lo, _ = st.step(x, y, noise)
        assert abs(float(res[0]) - lo) <= 5e-5 * abs(lo), (k, float(res[0]), lo)
    mu_o, rho_o = flat(st.mu, "inception"), flat(st.rho, "inception")
    # Adam normalises the update to ~lr per element: compare the displacement
    d_dev = eng.mu.cpu().double() - flat({k: v for k, v in R.init_mu0("inception", 0, torch.float64).items()}, "inception")
    d_orc = mu_o - flat(R.init_mu0("inception", 0, torch.float64), "inception")
    assert rel_l2(d_dev, d_orc) < 2e-2, rel_l2(d_dev, d_orc)
    assert rel_l2(eng.rho.cpu(), rho_o) < 1e-4


@pytest.mark.parametrize("mode", ["lrt", "flipout", "radial"])
def test_philox_noise_replayed_in_oracle(mode):
    """RNG mode: the kernels' own Philox noise, exported and replayed through the oracle."""
    S, B = 2, 5
    eng, cfg, st, x, y, _, (ps, qs, lr) = _setup("inception", mode, "f32", S, B, q_boost=20.0)
    res = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, seed=77, step=3)
    inj = eng.export_noise(B, S, seed=77, step=3)
    noise = from_injected(eng, cfg, inj, B, S)
    loss_o, aux = st.loss_and_grads(x, y, noise)
    assert abs(float(res[0]) - float(loss_o)) <= 5e-5 * abs(float(loss_o)), (float(res[0]), float(loss_o))
    g = eng.grad.cpu()
    gmu = torch.cat([st.mu[s].grad.flatten() for s, _ in R.site_shapes("inception")])
    assert rel_l2(g[:eng.P], gmu) < 5e-4
    # the exported normals look normal
    e = inj.eps_w.flatten().cpu()
    assert abs(float(e.mean())) < 0.01 and abs(float(e.std()) - 1) < 0.01
    if mode == "flipout":
        s = torch.cat([t.flatten() for t in inj.sign_in]).cpu()
        assert set(s.unique().tolist()) == {-1.0, 1.0} and abs(float(s.mean())) < 0.02


def test_validation_and_kl_only_losses():
    """evaluate_loss outside fit_ctxt (plain sampling, bayesian.py:177) and svi_no_obs (:155)."""
    S, B = 2, 6
    eng, cfg, st, x, y, _, (ps, qs, lr) = _setup("inception", "lrt", "f32", S, B, q_boost=20.0)
    from bayesrul_amd import _native as N
    noise = R.make_noise(cfg, B, S, torch.Generator().manual_seed(5), mode="normal")
    inj = to_injected(eng, cfg, noise, B, mode="normal")
    res, preds = eng.evaluate(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, mode=N.MODE_NORMAL, noise=inj, want_preds=True)
    lo, aux = st.evaluate_loss(x, y, noise, mode="plain")
    assert abs(float(res[0]) - lo) <= 2e-5 * abs(lo)
    assert torch.allclose(preds.cpu().double(), aux["preds"], rtol=2e-4, atol=1e-5)
    res = eng.evaluate(None, None, S, N_DATA, 0.0, ps, with_obs=False, scaled=False)
    lo, aux = st.evaluate_loss(x, y, noise, mode="plain", obs=False, scaled=False)
    assert abs(float(res[0]) - lo) <= 1e-5 * abs(lo)


@pytest.mark.parametrize("guide", ["normal", "radial"])
def test_predictive_pass_matches_oracle(guide):
    """tasks.predict path (A16): S plain-sampled forwards + ep/al variance aggregation, with the
    particles walked in chunks (max_windows < S*B)."""
    S, B = 5, 11
    mode = "radial" if guide == "radial" else "flipout"
    ps, qs, lr = HYP[mode]
    from bayesrul_amd.engine import SviEngine
    eng = SviEngine(net="inception", guide=guide, fit_context="flipout" if guide == "normal" else None, prec="f32",
                    max_particles=S, max_batch=B, max_windows=2 * B)
    mu0 = R.init_mu0("inception", 0, torch.float64)
    eng.init_params(mu0, qs * 50)
    cfg = oracle_cfg("inception", "radial" if guide == "radial" else "normal", ps)
    st = R.SviState(cfg, mu0, qs * 50, R.AdamConfig())
    x, y = synth_batch(B)
    noise = R.make_noise(cfg, B, S, torch.Generator().manual_seed(9))
    inj = to_injected(eng, cfg, noise, B)
    out4, samples = eng.predict(x.cuda(), S, noise=inj)
    _, aux = st.evaluate_loss(x, y, noise, mode="plain")
    ref = R.predictive_aggregate(aux["preds"])
    assert torch.allclose(samples.cpu().double(), aux["preds"], rtol=2e-4, atol=1e-5)
    o = out4.cpu().double()
    assert torch.allclose(o[0], ref["preds"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(o[1], ref["stds"], rtol=1e-3, atol=1e-5)
    assert torch.allclose(o[2], ref["ep_vars"], rtol=5e-3, atol=1e-7)
    assert torch.allclose(o[3], ref["al_vars"], rtol=1e-4, atol=1e-6)


def test_predict_and_validation_bf16x3_agree_with_f32():
    """The bf16x3 kernels on the no-gradient paths (plain sampling: tasks.predict and validation_step): same Philox
    noise through an f32 and a bf16x3 engine; predict walks the particles in chunks of two."""
    S, B = 6, 70
    ps, qs, lr = HYP["flipout"]
    from bayesrul_amd import _native as N
    from bayesrul_amd.engine import SviEngine
    mu0 = R.init_mu0("inception", 0, torch.float64)
    x, y = synth_batch(B)
    out = {}
    for prec in ("f32", "bf16x3"):
        eng = SviEngine(net="inception", guide="normal", fit_context="flipout", prec=prec, max_particles=S, max_batch=B,
                        max_windows=2 * B)   # predict walks the particles in chunks (the Philox stream depends on the chunking)
        eng.init_params(mu0, qs * 20)
        o4, samples = eng.predict(x.cuda(), S, seed=5)
        eng = SviEngine(net="inception", guide="normal", fit_context="flipout", prec=prec, max_particles=S, max_batch=B)
        eng.init_params(mu0, qs * 20)
        res = eng.evaluate(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, mode=N.MODE_NORMAL, seed=7)
        out[prec] = (o4.cpu().double(), samples.cpu().double(), float(res[0]))
    a, b = out["f32"], out["bf16x3"]
    assert torch.allclose(b[1], a[1], rtol=2e-3, atol=1e-4)          # per-particle predictions
    assert torch.allclose(b[0][0], a[0][0], rtol=1e-3, atol=1e-4)    # aggregated mean
    assert torch.allclose(b[0][1], a[0][1], rtol=5e-3, atol=1e-4)    # total std
    assert abs(b[2] - a[2]) <= 2e-4 * abs(a[2])                       # validation ELBO


def test_ragged_and_single_window_batches():
    """B = 1 (one window) and B = 33 (dense chunk of 32 + 1) keep parity (edge cases)."""
    for B in (1, 33):
        S = 1
        eng, cfg, st, x, y, noise, (ps, qs, lr) = _setup("inception", "lrt", "f32", S, B, q_boost=20.0)
        inj = to_injected(eng, cfg, noise, B)
        res = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj)
        loss_o, aux = st.loss_and_grads(x, y, noise)
        assert abs(float(res[0]) - float(loss_o)) <= 2e-5 * abs(float(loss_o)), B
        g = eng.grad.cpu()
        gmu = torch.cat([st.mu[s].grad.flatten() for s, _ in R.site_shapes("inception")])
        assert rel_l2(g[:eng.P], gmu) < 2e-4, B


@pytest.mark.parametrize("mode,S,B", [("lrt", 1, 300), ("flipout", 2, 257), ("radial", 3, 97), ("flipout", 8, 41)])
def test_fp32_dense_scheduling_geometries(mode, S, B):
    """The wide dense layer on the exact-fp32 plan: workgroups take balanced (particle, chunk, row step) item ranges that cross
    pair boundaries (forward, dX), and the dW splits a pair's rows into up to 8 ranges whose partial images one launch adds in
    order (S = 1, B = 300: 8 ranges; S = 2: 8 ranges x 2 particles; S = 3: 4 row steps; S = 8: 2 steps, 3 ranges).  Every site's
    gradient against the f64 oracle, so a mis-addressed partial image or a row step counted twice cannot hide."""
    eng, cfg, st, x, y, noise, (ps, qs, lr) = _setup("inception", mode, "f32", S, B, q_boost=20.0)
    inj = to_injected(eng, cfg, noise, B)
    res = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj)
    loss_o, aux = st.loss_and_grads(x, y, noise)
    assert abs(float(res[0]) - float(loss_o)) <= 2e-5 * abs(float(loss_o)), (float(res[0]), float(loss_o))
    g = eng.grad.cpu()
    for s, off, num in eng.sites:
        assert rel_l2(g[off:off + num], st.mu[s].grad) < 1e-3, ("mu", s)
        assert rel_l2(g[eng.P + off:eng.P + off + num], st.rho[s].grad) < 1e-3, ("rho", s)


@pytest.mark.parametrize("mode", ["flipout", "radial"])
def test_ragged_batches_bf16x3(mode):
    """The role-specialised bf16x3 kernels on awkward geometries: one window, a dense chunk of 32 + 1, more
    window sets than windows per particle, a partial last dense window, and enough particles for the store-only dense dW
    path (one workgroup per (particle, chunk), no gradient-image fill) with a ragged window - ELBO within the north star's
    1e-3 (held: 2e-4) and gradients within the single-bf16 backward tolerance."""
    for S, B in ((1, 1), (3, 33), (2, 257), (8, 41)):
        eng, cfg, st, x, y, noise, (ps, qs, lr) = _setup("inception", mode, "bf16x3", S, B)
        inj = to_injected(eng, cfg, noise, B)
        res = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj)
        loss_o, aux = st.loss_and_grads(x, y, noise)
        assert abs(float(res[0]) - float(loss_o)) <= 2e-4 * abs(float(loss_o)), (S, B, float(res[0]), float(loss_o))
        g = eng.grad.cpu()
        gmu = torch.cat([st.mu[s].grad.flatten() for s, _ in R.site_shapes("inception")])
        assert rel_l2(g[:eng.P], gmu) < 3e-2, (S, B)


def test_full_size_properties():
    """BASELINE sizes (S=10, B=1000): size-independent properties instead of the oracle.
    (1) f32 and bf16x3 paths agree on the ELBO to 1e-4 on the same Philox noise;
    (2) the step is deterministic for a fixed (seed, step) up to fp32 atomics order;
    (3) KL is independent of the batch."""
    S, B = 10, 1000
    ps, qs, lr = HYP["flipout"]
    x, y = synth_batch(B)
    mu0 = R.init_mu0("inception", 0, torch.float64)
    out = {}
    for prec in ("f32", "bf16x3"):
        eng = _engine("inception", "flipout", prec, S, B)
        eng.init_params(mu0, qs)
        r1 = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, seed=1, step=0).cpu().double()
        g1 = eng.grad.cpu().clone()
        r2 = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, seed=1, step=0).cpu().double()
        g2 = eng.grad.cpu().clone()
        assert abs(float(r1[0] - r2[0])) <= 1e-6 * abs(float(r1[0]))
        assert rel_l2(g1[:eng.P], g2[:eng.P]) < 1e-4
        out[prec] = (r1, g1)
        del eng
        torch.cuda.empty_cache()
    a, b = out["f32"][0], out["bf16x3"][0]
    assert abs(float(a[0] - b[0])) <= 1e-4 * abs(float(a[0])), (a, b)
    assert abs(float(a[1] - b[1])) <= 1e-6 * abs(float(a[1]))
    P = R.n_params("inception")
    assert rel_l2(out["bf16x3"][1][:P], out["f32"][1][:P]) < 3e-2
    assert rel_l2(out["bf16x3"][1][P:2 * P], out["f32"][1][P:2 * P]) < 8e-2   # d/d rho (single-bf16 backward)


def test_dp_step_world1_equals_plain_step():
    """`parallel.dp_step` (what `bench.py --gpus N` runs per rank: local step with adam=None -> all-reduce of grad[2P+2]
    -> ClippedAdam with grad_scale) with world = 1 is the plain `eng.step(adam=hyp)`: parameters, Adam moments, the step
    counter and the decayed lr agree bit for bit over three steps, and the returned (loss, kl) come from the gradient
    buffer's tail."""
    from bayesrul_amd.engine import AdamHyper
    from bayesrul_amd.parallel import dp_step
    S, B = 2, 64
    ps, qs, lr = HYP["flipout"]
    x, y = synth_batch(B)
    xg, yg = x.cuda(), y.cuda()
    mu0 = R.init_mu0("inception", 0, torch.float64)
    hyp = AdamHyper(lr=lr, betas=(0.95, 0.999), clip_norm=15.0)
    ea = _engine("inception", "flipout", "bf16x3", S, B)
    eb = _engine("inception", "flipout", "bf16x3", S, B)
    ea.init_params(mu0, qs)
    eb.init_params(mu0, qs)
    for k in range(3):
        ra = ea.step(xg, yg, S, N_DATA, 0.0, ps, hyp, seed=3, keep=True)
        rb = dp_step(eb, xg, yg, S, N_DATA, 0.0, ps, hyp, rank=0, world=1, seed=3)
        assert torch.equal(ra[:2].cpu(), rb[:2].cpu()), (k, ra, rb)
        assert ea.t == eb.t and ea.lr == eb.lr
    assert torch.equal(ea.mu, eb.mu) and torch.equal(ea.rho, eb.rho)
    assert torch.equal(ea.adam_m, eb.adam_m) and torch.equal(ea.adam_v, eb.adam_v)


def test_dp_noise_is_rank_invariant():
    """The Philox noise of a shard [off, off+B) of a global batch equals the slice of the
    noise of the whole batch (per-window streams are indexed by the GLOBAL window id) and the
    weight-level noise is identical on every rank."""
    S, Bg, B = 2, 8, 4
    for mode in ("lrt", "flipout", "radial"):
        eng = _engine("inception", mode, "f32", S, Bg)
        full = eng.export_noise(Bg, S, seed=5, step=9)
        for off in (0, 4):
            part = eng.export_noise(B, S, seed=5, step=9, global_batch=Bg, global_batch_offset=off)
            assert torch.equal(part.eps_w, full.eps_w) and torch.equal(part.radial_r, full.radial_r)
            for li in range(eng.n_layers):
                assert torch.equal(part.lrt_eps[li], full.lrt_eps[li][:, off:off + B]), (mode, li)
                assert torch.equal(part.sign_in[li], full.sign_in[li][:, off:off + B])
                assert torch.equal(part.sign_out[li], full.sign_out[li][:, off:off + B])
        del eng


def test_dp_two_shards_equal_global_step():
    """1 GPU on the global batch == mean of two shard steps (what 2 ranks + all-reduce compute)."""
    S, Bg = 2, 8
    ps, qs, lr = HYP["flipout"]
    x, y = synth_batch(Bg)
    mu0 = R.init_mu0("inception", 0, torch.float64)
    eng = _engine("inception", "flipout", "f32", S, Bg)
    eng.init_params(mu0, qs * 20)
    eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, seed=3, step=1)
    g_full = eng.grad.cpu().clone()
    acc = torch.zeros_like(g_full)
    for off in (0, 4):
        eng.step(x[off:off + 4].contiguous().cuda(), y[off:off + 4].contiguous().cuda(), S, N_DATA, 0.0, ps, None,
                 seed=3, step=1, global_batch=Bg, global_batch_offset=off)
        acc += eng.grad.cpu()
    acc /= 2
    assert rel_l2(acc[:-2], g_full[:-2]) < 1e-5
    assert abs(float(acc[-2] - g_full[-2])) < 1e-5 * abs(float(g_full[-2]))
