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


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "deepens":
        deepens()
    else:
        main()
        deepens()
