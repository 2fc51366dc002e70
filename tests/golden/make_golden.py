"""Generates tests/golden/*.npz from the parts of the reference that import in the dev
container.  Run HERE only (`python tests/golden/make_golden.py`); the
reference never travels to the GPU box, the committed .npz fixtures do.

`torchinfo` is imported by the reference nets only for their `__main__` blocks
(nets/inception.py:7,227-229; nets/linear.py:5,75-77); a dummy module stands in for it.
`shapely` / `uncertainty_toolbox` are likewise only needed by unrelated functions of
results/metrics.py.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    _stub("torchinfo", summary=lambda *a, **k: None)
    sys.path.insert(0, REF)
    from bayesrul.models.nets.inception import Inception
    from bayesrul.models.nets.linear import Linear
    from bayesrul.utils.miscellaneous import weights_init

    g = torch.Generator().manual_seed(1234)
    x = torch.randn(8, 30, 18, generator=g)

    for tag, ctor in (("inception", lambda: Inception(30, 18)), ("linear", lambda: Linear(30, 18, out_size=2))):
        torch.manual_seed(0)
        net = ctor()
        net.apply(weights_init)
        net.eval()
        with torch.no_grad():
            y = net(x)
        sd = {k: v.numpy() for k, v in net.state_dict().items()}
        np.savez_compressed(os.path.join(HERE, f"ref_{tag}_forward.npz"), x=x.numpy(), y=y.numpy(),
                            **{"sd::" + k: v for k, v in sd.items()})
        # gradient of a scalar of the output wrt input-side weights pins max_pool tie-breaking etc.
        net.zero_grad()
        out = net(x)
        (out[:, 0].sum() + 2.0 * out[:, 1].sum()).backward()
        grads = {k: p.grad.numpy() for k, p in net.named_parameters()}
        np.savez_compressed(os.path.join(HERE, f"ref_{tag}_grads.npz"), **{"g::" + k: v for k, v in grads.items()})
        print(tag, "params", sum(v.size for v in sd.values()), "y[0]", y[0].tolist())

    # metrics that import with stubs
    try:
        _stub("shapely"); _stub("shapely.geometry", Polygon=object, LineString=object)
        _stub("shapely.ops", polygonize=None, unary_union=None)
        _stub("uncertainty_toolbox"); _stub("uncertainty_toolbox.metrics_calibration", get_proportion_lists_vectorized=None)
        from bayesrul.results.metrics import nasa_score, sharpness
        s = torch.rand(64, generator=g) + 0.1
        yt = torch.rand(64, generator=g) * 100
        yp = yt + torch.randn(64, generator=g) * 10
        np.savez(os.path.join(HERE, "ref_metrics.npz"), s=s.numpy(), yt=yt.numpy(), yp=yp.numpy(),
                 sharp=sharpness(s).numpy(), nasa=nasa_score(yt, yp).numpy())
        print("metrics ok")
    except Exception as e:  # pragma: no cover
        print("metrics fixtures skipped:", repr(e))
    rmsce_and_frequentist(g)


def rmsce_and_frequentist(g):
    """rms_calibration_error (results/metrics.py:216-274) and the frequentist siblings HNN / NN
    (models/frequentist.py:39-58,157-188): the reference's OWN code, run here on CPU.

    Two environment shims, neither touching reference code:
      * `Tensor.get_device()` is -1 for CPU tensors and the reference feeds it to `torch.linspace(device=...)`,
        which raises.  For the duration of the run it returns torch.device("cpu").
      * `pytorch_lightning` is absent: a stand-in module supplies `LightningModule` = nn.Module +
        `save_hyperparameters` (collects the caller's ctor arguments) + `log` (records the value).  Only the
        module's own step arithmetic is exercised; the Lightning loop is replaced by an explicit
        zero_grad / backward / optimizer.step.
    """
    import functools
    import inspect

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.logged = {}

        def save_hyperparameters(self, *a, logger=True, ignore=None):
            loc = inspect.currentframe().f_back.f_locals
            self.hparams = types.SimpleNamespace(
                **{k: v for k, v in loc.items() if k not in ("self", "__class__") and k not in (ignore or [])})

        def log(self, name, value, **kw):
            self.logged[name] = float(value)

    _stub("pytorch_lightning", LightningModule=LightningModule)
    orig = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: self.device
    try:
        from bayesrul.models.frequentist import HNN, NN
        from bayesrul.models.nets.inception import Inception
        from bayesrul.results.metrics import rms_calibration_error

        # ---- rmsce on three synthetic predictors (well calibrated, over-, under-confident)
        n = 512
        std = torch.rand(n, generator=g) + 0.5
        yt = torch.rand(n, generator=g) * 100
        yp = yt + torch.randn(n, generator=g) * std
        out = {"std": std.numpy(), "yt": yt.numpy(), "yp": yp.numpy()}
        for tag, f in (("cal", 1.0), ("over", 0.3), ("under", 3.0)):
            out["rmsce_" + tag] = rms_calibration_error(yp, std * f, yt).numpy()
        np.savez(os.path.join(HERE, "ref_rmsce.npz"), **out)
        print("rmsce", {k: float(v) for k, v in out.items() if k.startswith("rmsce")})

        # ---- HNN / NN: 3 optimiser steps at the shipped hyper-parameters (conf/experiment/ncmapss_hnn.yaml:17-23,
        # conf/model/nn.yaml:6-10)
        B = 16
        x = torch.randn(B, 30, 18, generator=g)
        y = torch.randint(0, 100, (B,), generator=g).float()
        for tag, make in (("hnn", lambda net: HNN(net, functools.partial(torch.optim.Adam, lr=0.001574, weight_decay=1e-3),
                                                   mc_samples=0, p_dropout=0)),
                          ("nn", lambda net: NN(net, functools.partial(torch.optim.Adam, lr=0.001, weight_decay=1e-3)))):
            torch.manual_seed(0)
            model = make(Inception(30, 18))   # the ctor applies weights_init (frequentist.py:29, :169)
            sd0 = {k: v.detach().clone().numpy() for k, v in model.net.state_dict().items()}
            opt = model.configure_optimizers()
            losses, logs = [], []
            for i in range(3):
                opt.zero_grad()
                loss = model.training_step((x, y), i)
                loss.backward()
                opt.step()
                losses.append(float(loss))
                logs.append(dict(model.logged))
            sd1 = {k: v.detach().clone().numpy() for k, v in model.net.state_dict().items()}
            extra = {}
            if tag == "hnn":
                extra = {"mse": np.array([l["mse/train"] for l in logs]), "rmsce": np.array([l["rmsce/train"] for l in logs]),
                         "sharp": np.array([l["sharp/train"] for l in logs])}
            np.savez_compressed(os.path.join(HERE, f"ref_{tag}_steps.npz"), x=x.numpy(), y=y.numpy(),
                                losses=np.array(losses), **extra, **{"sd0::" + k: v for k, v in sd0.items()},
                                **{"sd1::" + k: v for k, v in sd1.items()})
            print(tag, "losses", losses)
    finally:
        torch.Tensor.get_device = orig


def deepens():
    """ref_deepens.npz: bayesrul/models/deepens.py (imports unaided) on a synthetic prediction table: 2 methods x 5 models
    x 40 windows; `deep_ensemble` of one method's models, and the first ensembles `deep_ensemble_gen` draws."""
    import pandas as pd
    sys.path.insert(0, REF)
    from bayesrul.models.deepens import deep_ensemble, deep_ensemble_gen
    rng = np.random.default_rng(7)
    n, rows = 40, []
    labels = rng.uniform(0, 100, n)
    for method in ("HNN", "MCD"):
        for k in range(5):
            rows.append(pd.DataFrame({"method": method, "model": f"{method}_{k:03d}", "labels": labels,
                                      "preds": labels + rng.normal(0, 5, n), "stds": rng.uniform(1, 9, n)}))
    df = pd.concat(rows, ignore_index=True)
    one = deep_ensemble(df.query("method=='HNN'"))
    gens = list(deep_ensemble_gen(df, ["HNN", "MCD"], 3, 4))
    out = {"labels": labels, "preds": df.preds.values, "stds": df.stds.values, "one_preds": one.preds.values,
           "one_stds": one.stds.values, "gen_preds": np.stack([g.preds.values for g in gens]),
           "gen_stds": np.stack([g.stds.values for g in gens]), "gen_models": np.array([g.model.iloc[0] for g in gens])}
    np.savez_compressed(os.path.join(HERE, "ref_deepens.npz"), **out)
    print("deepens", one.stds.values[:3], [g.model.iloc[0] for g in gens])


def mcd():
    """ref_mcd.npz: the reference's MC-dropout sibling (HNN on Inception(dropout = p), conf/experiment/ncmapss_mcd.yaml:21-30)
    run HERE on CPU with the environment shims of rmsce_and_frequentist().  nn.Dropout draws its masks from torch's
    generator, which the device path does not reproduce; forward hooks on the nine Dropout modules record the masks each
    pass used (keep = output != 0 where the input is non-zero; where the input is zero the mask is immaterial: kept), and
    the fixture holds them in the layout bnn_det_step / bnn_det_forward take as injected masks:
      two training steps (dropout active, frequentist.py:50): masks, losses, weights after the steps;
      `mc_sampling(batch, 3, "val")` after them (frequentist.py:60-81, `enable_dropout`): masks of the three passes, the
      aggregated (loss, loc, scale)."""
    import functools
    import inspect

    class LightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.logged = {}

        def save_hyperparameters(self, *a, logger=True, ignore=None):
            loc = inspect.currentframe().f_back.f_locals
            self.hparams = types.SimpleNamespace(
                **{k: v for k, v in loc.items() if k not in ("self", "__class__") and k not in (ignore or [])})

        def log(self, name, value, **kw):
            self.logged[name] = float(value)

    _stub("torchinfo", summary=lambda *a, **k: None)
    _stub("shapely"); _stub("shapely.geometry", Polygon=object, LineString=object)
    _stub("shapely.ops", polygonize=None, unary_union=None)
    _stub("uncertainty_toolbox"); _stub("uncertainty_toolbox.metrics_calibration", get_proportion_lists_vectorized=None)
    _stub("pytorch_lightning", LightningModule=LightningModule)
    sys.path.insert(0, REF)
    orig = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: self.device
    try:
        from bayesrul.models.frequentist import HNN
        from bayesrul.models.nets.inception import Inception
        from bayesrul.utils.miscellaneous import enable_dropout
        p, B, L = 0.241437, 12, 30
        g = torch.Generator().manual_seed(99)
        x = torch.randn(B, L, 18, generator=g)
        y = torch.randint(0, 100, (B,), generator=g).float()
        torch.manual_seed(0)
        model = HNN(Inception(L, 18, dropout=p), functools.partial(torch.optim.Adam, lr=0.000772, weight_decay=1e-3),
                    mc_samples=3, p_dropout=p)
        sd0 = {k: v.detach().clone().numpy() for k, v in model.net.state_dict().items()}
        # hooks: module name -> (tensor of block, first channel in the device layout)
        where = {"layers.0.conv1.branch1_dropout": (1, 0), "layers.0.conv3.branch2_dropout": (1, 32),
                 "layers.0.conv5.branch3_dropout": (1, 64), "layers.0.convpool.branch4_dropout": (1, 96),
                 "layers.1.branch1.branch1_dropout": (2, 0), "layers.1.branch2.branch2_dropout": (2, 16),
                 "layers.1.branch3.branch3_dropout": (2, 32), "layers.1.branch4.branch4_dropout": (2, 48),
                 "layers.n-1_dropout": (3, 0)}
        cur = {}

        def hook(name):
            def f(mod, inp, out):
                keep = torch.where(inp[0] != 0, out != 0, torch.ones_like(out, dtype=torch.bool)).float()
                t, c0 = where[name]
                if t == 3:
                    cur["kh"] = keep.clone()
                else:
                    buf = cur.setdefault("k1" if t == 1 else "k2", torch.ones(B * L, 128 if t == 1 else 80))
                    kk = keep.permute(0, 2, 1).reshape(B * L, -1)      # [B, C, L] -> rows b * L + l
                    buf[:, c0:c0 + kk.shape[1]] = kk
            return f
        seen = 0
        for name, mod in model.net.named_modules():
            if mod.__class__.__name__.startswith("Dropout"):
                mod.register_forward_hook(hook(name))
                seen += 1
        assert seen == 9, seen
        out = {"x": x.numpy(), "y": y.numpy(), "p": np.float64(p)}
        opt = model.configure_optimizers()
        model.train()
        losses = []
        for i in range(2):
            cur.clear()
            opt.zero_grad()
            loss = model.training_step((x, y), i)
            loss.backward()
            opt.step()
            losses.append(float(loss))
            for k in ("k1", "k2", "kh"):
                out[f"train{i}_{k}"] = cur[k].numpy().copy()
        out["train_losses"] = np.array(losses)
        sd1 = {k: v.detach().clone().numpy() for k, v in model.net.state_dict().items()}
        model.eval()
        enable_dropout(model.net)
        passes = []
        orig_step = model.step

        def step(batch, phase):
            cur.clear()
            r = orig_step(batch, phase)
            passes.append({k: cur[k].numpy().copy() for k in ("k1", "k2", "kh")})
            return r
        model.step = step
        with torch.no_grad():
            vloss, vloc, vscale = model.mc_sampling((x, y), 3, phase="val")
        for j, m in enumerate(passes):
            for k, v in m.items():
                out[f"val{j}_{k}"] = v
        out.update(val_loss=np.float64(float(vloss)), val_loc=vloc.numpy(), val_scale=vscale.numpy())
        np.savez_compressed(os.path.join(HERE, "ref_mcd.npz"), **out, **{"sd0::" + k: v for k, v in sd0.items()},
                            **{"sd1::" + k: v for k, v in sd1.items()})
        print("mcd losses", losses, "val", float(vloss), "kept fraction block1", float(out["train0_k1"][:, :27].mean()))
    finally:
        torch.Tensor.get_device = orig


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "deepens":
        deepens()
    elif len(sys.argv) > 1 and sys.argv[1] == "mcd":
        mcd()
    else:
        main()
        deepens()
        mcd()
