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
