"""The drop-in surface (`BNN` LightningModule-shaped class, guides, Trainer loops) on the GPU."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(guide="normal", ctx="lrt", S=1, prec="bf16x3"):
    from bayesrul_amd.models.bayesian import BNN
    from bayesrul_amd.models.nets.inception import Inception
    torch.manual_seed(0)
    net = Inception(30, 18)
    return BNN(net, {"lr": 2e-3, "betas": [0.95, 0.999], "clip_norm": 15}, pretrain_epochs=5, mc_samples_train=S,
               mc_samples_eval=4, dataset_size=512, fit_context=ctx, prior_loc=0.0, prior_scale=0.14, guide=guide,
               q_scale=0.0014, prec=prec, max_batch=64, max_eval_batch=128)


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


@pytest.mark.parametrize("kind,prec,tol", [("hnn", "f32", 2e-4), ("nn", "f32", 2e-4), ("hnn", "bf16x3", 5e-2)])
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
    for a, b in zip(losses, ref):
        assert abs(a - b) <= max(tol, 5 * tol if prec != "f32" else tol) * abs(b), (losses, ref.tolist())
    if kind == "hnn":
        logs = model.collect_logs()
        assert abs(logs["mse/train"] - float(z["mse"].mean())) <= 10 * tol * float(z["mse"].mean())
    sd = model.sync_net().state_dict()
    for k in sd:
        if k.startswith(("layers", "last")):
            d0, d1 = torch.from_numpy(z["sd0::" + k]), torch.from_numpy(z["sd1::" + k])
            # Adam moves every element by ~lr per step: compare the displacement
            num, den = (sd[k].cpu() - d1).norm(), (d1 - d0).norm()
            assert float(num) <= (0.02 if prec == "f32" else 0.35) * float(den), (k, float(num), float(den))


def test_gather_windows_bounds():
    from bayesrul_amd.data.window_store import DeviceWindowStore
    x = torch.arange(5 * 30 * 18, dtype=torch.float32).view(5, 30, 18)
    st = DeviceWindowStore(x, torch.arange(5, dtype=torch.float32), batch_size=4, shuffle=False)
    xo, yo = st.gather(torch.tensor([4, 5, -1, 0]))
    assert torch.equal(xo[0].cpu(), x[4]) and torch.equal(xo[3].cpu(), x[0])
    assert torch.isnan(xo[1]).all() and torch.isnan(xo[2]).all() and torch.isnan(yo[1:3]).all()
