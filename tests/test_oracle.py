"""Pins the oracle (oracle/restatement.py) against everything the reference can pin here:
golden vectors generated from the reference's own importable files (tests/golden/make_golden.py)
and closed-form / distributional identities for the third-party arithmetic (parity unpinned)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R


def _load_sd(path):
    z = np.load(path)
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return z, sd


@pytest.mark.parametrize("net", ["inception", "linear"])
def test_deterministic_forward_matches_reference(net, golden_dir):
    z, sd = _load_sd(os.path.join(golden_dir, f"ref_{net}_forward.npz"))
    # state_dict key names / shapes
    assert [k for k, _ in R.site_shapes(net)] == list(sd.keys())
    for k, shp in R.site_shapes(net):
        assert tuple(sd[k].shape) == shp
    assert R.n_params(net) == sum(v.numel() for v in sd.values())
    x = torch.from_numpy(z["x"])

    def layer(name, kind, h, pad):
        return R.layer_plain(kind, h, sd[name + ".weight"], sd[name + ".bias"], pad)

    y = R.net_forward(net, x, layer)
    np.testing.assert_allclose(y.numpy(), z["y"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("net", ["inception", "linear"])
def test_backward_matches_reference(net, golden_dir):
    z, sd = _load_sd(os.path.join(golden_dir, f"ref_{net}_forward.npz"))
    g = np.load(os.path.join(golden_dir, f"ref_{net}_grads.npz"))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(z["x"])

    def layer(name, kind, h, pad):
        return R.layer_plain(kind, h, p[name + ".weight"], p[name + ".bias"], pad)

    out = R.net_forward(net, x, layer)
    (out[:, 0].sum() + 2.0 * out[:, 1].sum()).backward()
    for k in p:
        np.testing.assert_allclose(p[k].grad.numpy(), g["g::" + k], rtol=2e-4, atol=1e-6)


def test_metrics_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_metrics.npz"))
    s, yt, yp = (torch.from_numpy(z[k]) for k in ("s", "yt", "yp"))
    np.testing.assert_allclose(R.sharpness(s).numpy(), z["sharp"], rtol=1e-6)
    np.testing.assert_allclose(R.nasa_score(yt, yp).numpy(), z["nasa"], rtol=1e-6)


def test_rmsce_perfectly_calibrated_is_small():
    g = torch.Generator().manual_seed(0)
    n = 200000
    std = torch.rand(n, generator=g, dtype=torch.float64) + 0.5
    yt = torch.zeros(n, dtype=torch.float64)
    yp = torch.randn(n, generator=g, dtype=torch.float64) * std
    assert float(R.rms_calibration_error(yp, std, yt)) < 5e-3
    # over-confident predictor is badly calibrated
    assert float(R.rms_calibration_error(yp, std * 0.2, yt)) > 0.2


def test_kl_matches_torch_distributions():
    g = torch.Generator().manual_seed(1)
    mu = torch.randn(50, generator=g, dtype=torch.float64)
    rho = torch.randn(50, generator=g, dtype=torch.float64) * 0.3 - 2
    q = torch.distributions.Normal(mu, rho.exp())
    p = torch.distributions.Normal(torch.tensor(0.1, dtype=torch.float64), torch.tensor(0.3, dtype=torch.float64))
    ref = torch.distributions.kl_divergence(q, p).sum()
    assert torch.allclose(R.kl_normal_normal(mu, rho, 0.1, 0.3), ref, rtol=1e-12)


def test_radial_sample_properties():
    g = torch.Generator().manual_seed(2)
    mu = torch.randn(7, 5, generator=g, dtype=torch.float64)
    rho = torch.full_like(mu, math.log(0.2))
    eps = torch.randn(7, 5, generator=g, dtype=torch.float64)
    r = torch.tensor([1.7], dtype=torch.float64)
    w = R.sample_radial(mu, rho, eps, r)
    # ||(w - mu)/sigma||_2 == |r|  (guides/radial.py:37-41)
    assert torch.allclose(((w - mu) / 0.2).norm(), r.abs())
    # log q of the radial sample under the plain Normal log_prob (A6)
    lq = R.normal_log_prob(w, mu, rho.exp()).sum()
    n = mu.numel()
    expect = -0.5 * float(r) ** 2 - n * math.log(0.2) - n * 0.5 * math.log(2 * math.pi)
    assert abs(float(lq) - expect) < 1e-9


def test_lrt_moments_match_plain_sampling():
    """LRT output distribution == distribution of the plainly sampled layer (Kingma 2015)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 6, 10, generator=g, dtype=torch.float64)
    mu_w = torch.randn(5, 6, 3, generator=g, dtype=torch.float64) * 0.3
    rho_w = torch.full_like(mu_w, math.log(0.1))
    mu_b = torch.randn(5, generator=g, dtype=torch.float64)
    rho_b = torch.full_like(mu_b, math.log(0.05))
    n = 20000
    acc = torch.zeros(4, 5, 10, dtype=torch.float64)
    acc2 = torch.zeros_like(acc)
    for _ in range(n):
        w = R.sample_normal(mu_w, rho_w, torch.randn(mu_w.shape, generator=g, dtype=torch.float64))
        b = R.sample_normal(mu_b, rho_b, torch.randn(mu_b.shape, generator=g, dtype=torch.float64))
        o = R.layer_plain("conv", x, w, b, 1)
        acc += o
        acc2 += o * o
    mean, var = acc / n, acc2 / n - (acc / n) ** 2
    loc = R.layer_lrt("conv", x, mu_w, rho_w, mu_b, rho_b, torch.zeros(4, 5, 10, dtype=torch.float64), 1)
    out1 = R.layer_lrt("conv", x, mu_w, rho_w, mu_b, rho_b, torch.ones(4, 5, 10, dtype=torch.float64), 1)
    lrt_var = (out1 - loc) ** 2
    assert torch.allclose(mean, loc, atol=4 * float(var.max().sqrt()) / math.sqrt(n) + 1e-3)
    assert torch.allclose(var, lrt_var, rtol=0.08)


def test_flipout_marginal_matches_plain_sampling():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 8, generator=g, dtype=torch.float64)
    mu_w = torch.randn(4, 8, generator=g, dtype=torch.float64) * 0.3
    rho_w = torch.full_like(mu_w, math.log(0.2))
    b = torch.randn(4, generator=g, dtype=torch.float64)
    n = 20000
    acc = torch.zeros(3, 4, dtype=torch.float64)
    acc2 = torch.zeros_like(acc)
    for _ in range(n):
        w = R.sample_normal(mu_w, rho_w, torch.randn(mu_w.shape, generator=g, dtype=torch.float64))
        si = (torch.rand(3, 8, generator=g) > 0.5).double() * 2 - 1
        so = (torch.rand(3, 4, generator=g) > 0.5).double() * 2 - 1
        o = R.layer_flipout("linear", x, mu_w, w, b, si, so, 0)
        acc += o
        acc2 += o * o
    mean, var = acc / n, acc2 / n - (acc / n) ** 2
    loc = x @ mu_w.T + b
    exp_var = (x * x) @ (rho_w.exp() ** 2).T
    assert torch.allclose(mean, loc, atol=0.03)
    assert torch.allclose(var, exp_var, rtol=0.08)


def test_clipped_adam_equals_torch_adam_with_elementwise_clamp():
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(30, generator=g, dtype=torch.float64)
    ac = R.AdamConfig(lr=1e-2, beta1=0.95, beta2=0.999, clip_norm=0.5)
    p = p0.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-2, betas=(0.95, 0.999), eps=1e-8)
    for t in range(1, 6):
        grad = torch.randn(30, generator=g, dtype=torch.float64) * 2
        R.clipped_adam_step(p, grad, m, v, t, ac.lr, ac)
        pt.grad = grad.clamp(-0.5, 0.5)
        opt.step()
    # identical up to where eps enters (pyro: sqrt(v)+eps before bias correction)
    assert torch.allclose(p, pt.detach(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("mode", ["lrt", "flipout", "radial", "normal"])
def test_elbo_loss_runs_and_is_finite(mode):
    cfg = R.ElboConfig(net="inception", guide="radial" if mode == "radial" else "normal",
                       fit_context=mode if mode in ("lrt", "flipout") else None,
                       prior_scale=0.14)
    mu0 = R.init_mu0("inception", 0, torch.float64)
    st = R.SviState(cfg, mu0, 0.0013, R.AdamConfig(lr=8.57e-4))
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(5, 30, 18, generator=g, dtype=torch.float64)
    y = torch.randint(0, 100, (5,), generator=g).double()
    noise = R.make_noise(cfg, 5, 2, torch.Generator().manual_seed(4321))
    l0, aux = st.step(x, y, noise)
    assert math.isfinite(l0) and aux["preds"].shape == (2, 5, 2)
    l1, _ = st.step(x, y, noise)
    assert math.isfinite(l1)


def test_init_mu0_statistics():
    mu0 = R.init_mu0("inception", 0)
    w = mu0["layers.3.weight"]
    assert abs(float(w.std()) - math.sqrt(2 / 2400)) < 2e-3            # kaiming_normal_, fan_in
    w = mu0["layers.1.branch2.0.weight"]
    assert abs(float(w.std()) - math.sqrt(2 / (108 + 64))) < 5e-3      # xavier_normal_
