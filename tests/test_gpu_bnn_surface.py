"""The drop-in surface (`BNN` LightningModule-shaped class, guides, Trainer loops) on the GPU."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu



def _model(guide="normal", ctx="lrt", S=1, prec="auto", seed=0):
    from bayesrul_amd.models.bayesian import BNN
    from bayesrul_amd.models.nets.inception import Inception
    torch.manual_seed(0)
    net = Inception(30, 18)
    return BNN(net, {"lr": 2e-3, "betas": [0.95, 0.999], "clip_norm": 15}, pretrain_epochs=5, mc_samples_train=S,
               mc_samples_eval=4, dataset_size=512, fit_context=ctx, prior_loc=0.0, prior_scale=0.14, guide=guide,
               q_scale=0.0014, prec=prec, max_batch=64, max_eval_batch=128, seed=seed)


@pytest.mark.parametrize("guide,ctx,S", [("normal", "lrt", 1), ("normal", "flipout", 2), ("radial", None, 1)])
def test_fit_validate_test_predict(guide, ctx, S, tmp_path):
    from bayesrul_amd.data.synthetic import SyntheticWindows
    from bayesrul_amd.lightning_lite import Trainer
    model = _model(guide, ctx, S)
    train = SyntheticWindows(512, 64, shuffle=True, learnable=True)
    val = SyntheticWindows(128, 128, seed=7, learnable=True)
    tr = Trainer(max_epochs=6)
    hist = tr.fit(model, train, val)
    keys = {"mse/train", "elbo/train", "kl/train", "likelihood/train", "rmsce/train", "sharp/train", "elbo/val",
            "mse/val", "kl/val", "likelihood/val", "rmsce/val", "sharp/val"}
    assert keys <= set(hist[0])
    assert all(math.isfinite(v) for v in hist[-1].values())
    assert hist[-1]["elbo/train"] < hist[0]["elbo/train"]          # the ELBO loss goes down
    assert hist[-1]["mse/train"] < hist[0]["mse/train"]
    # guide surface
    g = model.bnn.net_guide
    assert g.get_loc("layers.3.weight").shape == (64, 2400)
    assert float(g.get_scale("last.bias").min()) > 0
    d = g.get_detached_distributions(["last.weight"])["last.weight"]
    assert d.rsample().shape == (2, 64)
    # checkpoint round trip through the Pyro-param-store shaped entry
    path = os.path.join(tmp_path, "m.ckpt")
    tr.save_checkpoint(model, path)
    model2 = _model(guide, ctx, S)
    tr2 = Trainer()
    ckpt = tr2.load_checkpoint(model2, path)
    assert "param_store" in ckpt and f"net_guide.last.bias.loc" in ckpt["param_store"]["params"]
    logs = tr2.test(model2, val)
    assert {"nll/test", "mse/test", "rmsce/test", "sharp/test"} <= set(logs)
    assert torch.equal(model2.engine.mu.cpu(), model.engine.mu.cpu())
    preds = tr2.predict(model2, val)
    assert set(preds[0]) == {"labels", "ep_vars", "al_vars", "preds", "stds"}
    assert preds[0]["preds"].shape == (128,) and np.all(preds[0]["ep_vars"] >= 0) and np.all(np.isfinite(preds[0]["stds"]))


@pytest.mark.parametrize("guide,ctx,S", [("normal", "flipout", 2), ("radial", None, 1)])
def test_200_step_fit_bf16x3_tracks_the_fp32_plan(guide, ctx, S):
    """does the split-bf16 plan's looser backward (single-bf16 gradient contractions) matter for a
    training run?  200 optimiser steps (25 epochs of 8) of the lite Trainer on the learnable synthetic set: the exact-fp32
    plan and the bf16x3 plan from the same initial weights, shuffles and Philox noise streams, and - as the yardstick of
    what a difference means for a stochastic training run - the exact-fp32 plan once more with another noise seed.
    MEASURED (MI355X, round 3): the validation ELBO of the two plans agrees to 4e-5 relative; MSE / calibration error /
    sharpness (4 MC samples) differ by 1.9 % / 0.028 / 8.6 % (Flipout, S = 2: within the seed-to-seed spread of 0.3 % / 0.071 /
    34 % except the MSE) and 3.4 % / 0.047 / 24 % (radial, S = 1: 2-4 x the seed-to-seed spread of 1.4 % / 0.025 / 6.4 %).
    So the bf16x3 plan IS an approximation that a 200-step run can see in its uncertainty metrics: it is opt-in, the
    exact-fp32 plan is the default of every class and of bench.py.  Bounds below = 2 x the measured differences."""
    from bayesrul_amd.data.synthetic import SyntheticWindows
    from bayesrul_amd.lightning_lite import Trainer
    from tests.noise_util import record
    final = {}
    for tag, prec, seed in (("f32", "f32", 0), ("bf16x3", "bf16x3", 0), ("f32_other_noise", "f32", 1000)):
        model = _model(guide, ctx, S, prec=prec, seed=seed)
        train = SyntheticWindows(512, 64, shuffle=True, learnable=True)
        val = SyntheticWindows(128, 128, seed=7, learnable=True)
        hist = Trainer(max_epochs=25).fit(model, train, val)
        final[tag] = hist[-1]
        assert hist[-1]["elbo/val"] < hist[0]["elbo/val"]
        del model
        torch.cuda.empty_cache()
    a, b, c = final["f32"], final["bf16x3"], final["f32_other_noise"]
    keys = ("elbo/val", "mse/val", "rmsce/val", "sharp/val")
    plan = {k: abs(a[k] - b[k]) for k in keys}     # plan-to-plan, same noise
    noise = {k: abs(a[k] - c[k]) for k in keys}    # same plan, other noise
    record(f"fit200[{guide}-{ctx}]", f32={k: a[k] for k in keys}, plan_diff=plan, noise_diff=noise)
    assert plan["elbo/val"] <= 1e-4 * abs(a["elbo/val"]), (a, b)   # measured 3e-5 .. 4e-5
    assert plan["mse/val"] <= 0.07 * abs(a["mse/val"]), (a, b)
    assert plan["rmsce/val"] <= 0.1, (a, b)
    assert plan["sharp/val"] <= 0.5 * abs(a["sharp/val"]), (a, b)


def test_lightning_progress_poke_and_resume(tmp_path):
    """bayesian.py:144,156: every training step increments the manual-optimisation progress tracker; a checkpoint loaded
    BEFORE the engine exists (load -> on_fit_start, the normal resume path) restores the Adam moments, the step count
    and the decayed learning rate, not only the param store."""
    from bayesrul_amd.data.synthetic import SyntheticWindows
    from bayesrul_amd.lightning_lite import Trainer
    model = _model("normal", "flipout", 2)
    train = SyntheticWindows(256, 64, shuffle=True, learnable=True)
    tr = Trainer(max_epochs=2)
    tr.fit(model, train)
    prog = tr.fit_loop.epoch_loop.batch_loop.manual_loop.optim_step_progress
    assert prog.ready == prog.completed == 8 == model.engine.t
    path = os.path.join(tmp_path, "m.ckpt")
    tr.save_checkpoint(model, path)
    model2 = _model("normal", "flipout", 2)
    tr2 = Trainer(max_epochs=1)
    tr2.load_checkpoint(model2, path)
    assert model2.engine is None
    model2.trainer = tr2
    model2.to(tr2.device)
    model2.on_fit_start()
    assert model2.engine.t == 8 and model2.engine.lr == model.engine.lr
    assert torch.equal(model2.engine.adam_m.cpu(), model.engine.adam_m.cpu())
    assert torch.equal(model2.engine.rho.cpu(), model.engine.rho.cpu())


def test_init_to_median_stand_in_and_guide_options():
    """pretrain_epochs == 0: means = per-element median of 15 prior draws from a generator seeded with `seed`
    (reproducible; std of the median of 15 normals = 0.3236 sigma); guide options: train_scale=False freezes rho,
    init_scale as a dict sets per-site scales, max_guide_scale is refused."""
    from bayesrul_amd.models.bayesian import BNN
    from bayesrul_amd.models.nets.inception import Inception
    from bayesrul_amd.data.synthetic import SyntheticWindows
    from bayesrul_amd.lightning_lite import Trainer

    def make(seed, **gk):
        torch.manual_seed(0)
        return BNN(Inception(30, 18), {"lr": 2e-3, "betas": [0.95, 0.999], "clip_norm": 15}, pretrain_epochs=0,
                   mc_samples_train=1, mc_samples_eval=2, dataset_size=256, fit_context="lrt", prior_loc=0.1,
                   prior_scale=0.2, guide="normal", q_scale=0.002, max_batch=64, max_eval_batch=64, seed=seed,
                   guide_kwargs=gk or None)

    ms = []
    for seed in (3, 3, 4):
        m = make(seed)
        m.to("cuda:0")
        m.on_fit_start()
        ms.append(m.engine.mu.cpu().clone())
    assert torch.equal(ms[0], ms[1]) and not torch.equal(ms[0], ms[2])
    assert abs(float(ms[0].mean()) - 0.1) < 2e-3 and abs(float(ms[0].std()) / 0.2 - 0.3236) < 0.01
    # frozen scales / per-site init
    scales = {n: torch.full((1,), 0.003 if n.endswith("bias") else 0.001) for n, _, _ in m.engine.sites}
    m = make(0, train_scale=False, init_scale=scales)
    tr = Trainer(max_epochs=1)
    tr.fit(m, SyntheticWindows(128, 64, learnable=True))
    g = m.bnn.net_guide
    assert torch.allclose(g.get_scale("last.bias"), torch.full((2,), 0.003, device="cuda:0"))
    assert torch.allclose(g.get_scale("layers.3.weight"), torch.full((64, 2400), 0.001, device="cuda:0"))
    assert float((m.engine.mu.cpu() - ms[0]).abs().max()) > 0    # the means did move
    with pytest.raises(RuntimeError):
        bad = make(0, max_guide_scale=0.1)
        bad.to("cuda:0")
        bad.on_fit_start()


def test_import_pyro_shaped_param_store():
    """A param store as the reference writes it (unconstrained tensors, names prefixed by TyXe's module nesting —
    the prefix is unverified, U12, so sites are matched by suffix)."""
    model = _model("radial", None, 1)
    model.to("cuda:0")
    model.on_fit_start()
    g = torch.Generator().manual_seed(1)
    params = {}
    for name, _, num in model.engine.sites:
        shp = model.engine.site_shape(name)
        params[f"net_guide.net.{name}.loc"] = torch.randn(shp, generator=g)
        params[f"net_guide.net.{name}.scale"] = torch.randn(shp, generator=g) - 6.0
    model.import_pyro_param_store({"params": params, "constraints": {}})
    assert torch.equal(model.engine.loc("layers.1.branch3.2.weight").cpu(), params["net_guide.net.layers.1.branch3.2.weight.loc"])
    assert torch.equal(model.engine.log_scale("last.bias").cpu(), params["net_guide.net.last.bias.scale"])


@pytest.mark.parametrize("kind,prec,tol", [("hnn", "f32", 2e-5), ("nn", "f32", 2e-5), ("hnn", "bf16x3", 1e-3)])
def test_frequentist_siblings_match_reference_steps(kind, prec, tol, golden_dir):
    """HNN / NN (bayesrul/models/frequentist.py:39-58,157-188) on the device kernels against three optimiser steps of
    the reference's own classes (tests/golden/make_golden.py: ref_{hnn,nn}_steps.npz; torch.optim.Adam with weight
    decay at the shipped hyper-parameters): losses and final weights.  The f32 plan is held to fp32 round-off; the
    bf16x3 plan (single-bf16 backward) to its gradient tolerance."""
    import functools
    from bayesrul_amd.models.frequentist import HNN, NN
    from bayesrul_amd.models.nets.inception import Inception
    z = np.load(os.path.join(golden_dir, f"ref_{kind}_steps.npz"))
    net = Inception(30, 18)
    if kind == "hnn":
        model = HNN(net, functools.partial(torch.optim.Adam, lr=0.001574, weight_decay=1e-3), mc_samples=0, p_dropout=0,
                    prec=prec, max_batch=16)
    else:
        model = NN(net, {"lr": 0.001, "weight_decay": 1e-3}, prec=prec, max_batch=16)
    net.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0::")})
    model.to("cuda:0")
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    losses = []
    for i in range(3):
        losses.append(float(model.training_step((x, y), i)))
    ref = z["losses"]
    from tests.noise_util import record
    record(f"frequentist[{kind}-{prec}]", loss_rel=[abs(a - b) / abs(b) for a, b in zip(losses, ref)])
    # bf16x3: the first loss is evaluated on identical weights (forward error only: measured 7e-6); the later ones after
    # updates driven by single-bf16 gradients (measured 4.3e-4).  f32: measured <= 1e-5 at every step.
    for i, (a, b) in enumerate(zip(losses, ref)):
        bound = tol if prec == "f32" else (2e-5 if i == 0 else 1e-3)
        assert abs(a - b) <= bound * abs(b), (i, losses, ref.tolist())
    if kind == "hnn":
        logs = model.collect_logs()
        assert abs(logs["mse/train"] - float(z["mse"].mean())) <= 1e-3 * float(z["mse"].mean())
    sd = model.sync_net().state_dict()
    disp, tot_num, tot_den = {}, 0.0, 0.0
    for k in sd:
        if k.startswith(("layers", "last")):
            d0, d1 = torch.from_numpy(z["sd0::" + k]), torch.from_numpy(z["sd1::" + k])
            # Adam moves every element by ~lr per step: compare the displacement
            num, den = (sd[k].cpu() - d1).norm(), (d1 - d0).norm()
            disp[k] = float(num) / float(den)
            # f32: measured 1e-5.  bf16x3: Adam moves every element by ~lr per step whatever the size of its gradient, so an
            # element whose gradient is within the bf16 backward's noise of zero moves the OTHER way: on a 16-element bias site
            # two such elements are 0.29 of the displacement's norm (measured worst site, layers.1.branch3.2.bias); the
            # whole-vector figure is asserted below
            assert float(num) <= (1e-4 if prec == "f32" else 0.35) * float(den), (k, float(num), float(den))
            tot_num += float(num) ** 2
            tot_den += float(den) ** 2
    glob = (tot_num / tot_den) ** 0.5
    record(f"frequentist_displacement_global[{kind}-{prec}]", rel_l2=glob)
    assert glob <= (1e-4 if prec == "f32" else 0.042), glob   # measured 7.9e-6 (f32), 2.1e-2 (bf16x3)
    record(f"frequentist_displacement[{kind}-{prec}]", worst=max(disp.values()), site=max(disp, key=disp.get))


def test_gather_windows_bounds():
    from bayesrul_amd.data.window_store import DeviceWindowStore
    x = torch.arange(5 * 30 * 18, dtype=torch.float32).view(5, 30, 18)
    st = DeviceWindowStore(x, torch.arange(5, dtype=torch.float32), batch_size=4, shuffle=False)
    xo, yo = st.gather(torch.tensor([4, 5, -1, 0]))
    assert torch.equal(xo[0].cpu(), x[4]) and torch.equal(xo[3].cpu(), x[0])
    assert torch.isnan(xo[1]).all() and torch.isnan(xo[2]).all() and torch.isnan(yo[1:3]).all()


def test_mc_dropout_matches_reference_with_injected_masks(golden_dir):
    """The MC-dropout sibling (frequentist.py:50-92 on Inception(dropout = p), conf/experiment/ncmapss_mcd.yaml) against the
    REFERENCE's own classes (tests/golden/make_golden.py mcd(): ref_mcd.npz).  nn.Dropout's masks come from torch's
    generator there; the fixture holds the masks every pass used and the device path takes them as injected keep masks:
    two training steps (dropout active) - losses and weights - then the three stochastic passes of
    `mc_sampling(batch, 3, "val")` - aggregated loss / loc / scale."""
    import torch.nn.functional as F
    from bayesrul_amd.models.frequentist import HNN
    from bayesrul_amd.models.nets.inception import Inception
    from tests.noise_util import record
    z = np.load(os.path.join(golden_dir, "ref_mcd.npz"))
    p = float(z["p"])
    net = Inception(30, 18, dropout=p)
    model = HNN(net, {"lr": 0.000772, "weight_decay": 1e-3}, mc_samples=3, p_dropout=p, prec="f32", max_batch=12)
    net.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0::")})
    model.to("cuda:0")
    x, y = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()
    eng = model._ensure_engine("f32", 12)
    keep = lambda tag: tuple(torch.from_numpy(z[f"{tag}_{k}"]) for k in ("k1", "k2", "kh"))
    losses = []
    for i in range(2):
        loss, _ = eng.det_step(x, y, "gaussian_nll", model.adam, dropout=eng._dropout(p, 0, 0, keep(f"train{i}")))
        losses.append(float(loss[0]))
    ref = z["train_losses"]
    record("mcd_train", loss_rel=[abs(a - b) / abs(b) for a, b in zip(losses, ref)])
    for a, b in zip(losses, ref):
        assert abs(a - b) <= 2e-5 * abs(b), (losses, ref.tolist())
    sd = model.sync_net().state_dict()
    tn = td = 0.0
    for k in sd:
        if k.startswith(("layers", "last")):
            d0, d1 = torch.from_numpy(z["sd0::" + k]), torch.from_numpy(z["sd1::" + k])
            tn += float((sd[k].cpu() - d1).norm()) ** 2
            td += float((d1 - d0).norm()) ** 2
    record("mcd_train_displacement", rel_l2=(tn / td) ** 0.5)
    assert (tn / td) ** 0.5 <= 1e-3, (tn / td) ** 0.5
    # mc_sampling(batch, 3, "val"): three stochastic passes with the recorded masks
    locs, scales, ls = [], [], []
    for j in range(3):
        out = eng.det_forward(x, eng._dropout(p, 0, 0, keep(f"val{j}")))
        locs.append(out[:, 0])
        scales.append(out[:, 1])
        ls.append(F.gaussian_nll_loss(out[:, 0], y, torch.square(out[:, 1])))
    locs, scales = torch.stack(locs), torch.stack(scales)
    loc, scale = locs.mean(0), scales.pow(2).mean(0).add(locs.var(0)).sqrt()
    record("mcd_val", loss_rel=abs(float(torch.stack(ls).mean()) - float(z["val_loss"])) / abs(float(z["val_loss"])),
           loc=float((loc.cpu() - torch.from_numpy(z["val_loc"])).abs().max()),
           scale=float((scale.cpu() - torch.from_numpy(z["val_scale"])).abs().max()))
    assert abs(float(torch.stack(ls).mean()) - float(z["val_loss"])) <= 2e-5 * abs(float(z["val_loss"]))
    assert torch.allclose(loc.cpu(), torch.from_numpy(z["val_loc"]), rtol=2e-5, atol=1e-6)
    assert torch.allclose(scale.cpu(), torch.from_numpy(z["val_scale"]), rtol=2e-5, atol=1e-6)


def test_mc_dropout_philox_masks_and_hooks():
    """The kernels' own masks (Philox): the share of block-2 outputs a stochastic pass zeroes beyond the ReLU zeros is the
    dropout rate p / 4 (nets/inception.py:119-123), passes differ, and the HNN hooks run the reference's branches
    (`mc_sampling` in validation / test / predict when net.dropout > 0)."""
    from bayesrul_amd import _native as N
    from bayesrul_amd.models.frequentist import HNN
    from bayesrul_amd.models.nets.inception import Inception
    torch.manual_seed(0)
    p = 0.4
    net = Inception(30, 18, dropout=p)
    model = HNN(net, {"lr": 1e-3, "weight_decay": 1e-3}, mc_samples=5, p_dropout=p, prec="f32", max_batch=256)
    model.to("cuda:0")
    x = torch.randn(256, 30, 18, device="cuda:0")
    y = torch.randint(0, 100, (256,), device="cuda:0").float()
    eng = model._ensure_engine("f32", 256)
    eng.det_forward(x, None)
    a0 = eng.tensor(N.T_ACT2)[:256 * 30]
    eng.det_forward(x, eng._dropout(p, 7, 1))
    a1 = eng.tensor(N.T_ACT2)[:256 * 30]
    # block-1 dropout changes the block-2 pre-activations too, so compare the zero SHARES, not positions
    z0, z1 = float((a0 == 0).float().mean()), float((a1 == 0).float().mean())
    expect = z0 + (1 - z0) * p / 4
    assert abs(z1 - expect) < 0.02, (z0, z1, expect)
    kept = a1[a1 != 0].abs().mean() / a0[a0 != 0].abs().mean()
    assert 0.8 < float(kept) < 1.4     # survivors are scaled by 1 / (1 - p / 4)
    o1, o2 = eng.det_forward(x, eng._dropout(p, 7, 1)), eng.det_forward(x, eng._dropout(p, 7, 2))
    assert torch.equal(o1, eng.det_forward(x, eng._dropout(p, 7, 1))) and not torch.equal(o1, o2)
    # hooks
    l0 = float(model.training_step((x, y), 0))
    out = model.validation_step((x, y), 0)
    assert out["pred"].shape == (256,) and torch.isfinite(out["std"]).all() and math.isfinite(l0)
    model.test_step((x, y), 0)
    pr = model.predict_step((x, y), 0)
    assert set(pr) == {"labels", "ep_vars", "al_vars", "preds", "stds"} and np.all(pr["ep_vars"] > 0)
    # refused where it cannot run
    with pytest.raises(RuntimeError):
        HNN(Inception(30, 18, dropout=p), {"lr": 1e-3}, mc_samples=2, p_dropout=p, prec="bf16x3").to("cuda:0").on_fit_start()
