"""CPU restatement of the reference's SVI/ELBO hot path (TEST INFRASTRUCTURE ONLY).

This module is the *oracle*: a plain-PyTorch CPU restatement of the arithmetic the
reference (lbasora/bayesrul) executes per training batch through Pyro/TyXe.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product path (``bayesrul_amd``) never does.

PARITY UNPINNED: the reference ships no tests / golden vectors for this path and the
third-party modules that hold the arithmetic (TyXe @368bf62, pyro-ppl 1.8.1) are not
present in ``/root/reference`` nor installable here.  What *is*
pinned (tests/golden/, generated from the reference's own importable files): the
deterministic ``Inception`` / ``Linear`` forward, state_dict key names and shapes,
``weights_init`` statistics, ``sharpness`` and ``nasa_score``.  The variational
arithmetic below follows the call sites in the reference and the published
algorithms of the pinned third-party versions; every assumption is tagged U1..U12 as
in SURVEY.md §8(c).

All randomness enters as explicit noise tensors so that any backend can be fed
identical noise.

Reference call sites restated here
  bayesrul/models/bayesian.py:45-98    define_bnn  (prior, fit context, likelihood, guide)
  bayesrul/models/bayesian.py:100-132  on_fit_start (ELBO choice, 1/(N*W*F) scaling)
  bayesrul/models/bayesian.py:134-166  training_step
  bayesrul/models/bayesian.py:203-250  test_step / predict_step aggregation
  bayesrul/models/guides/radial.py:31-41   RadialNormal.rsample
  bayesrul/models/nets/inception.py:10-217 Inception
  bayesrul/models/nets/linear.py:10-72     Linear
  bayesrul/results/metrics.py:210-274      sharpness, rms_calibration_error
  bayesrul/conf/model/bnn.yaml:6-10        ClippedAdam hyper-parameters
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Network descriptions (bayesrul/models/nets/inception.py:142-217, nets/linear.py:10-72)
# --------------------------------------------------------------------------------------
# (layer name, kind, cout, cin, kernel)
INCEPTION_LAYERS: List[Tuple[str, str, int, int, int]] = [
    ("layers.0.conv1.0", "conv", 27, 18, 1),
    ("layers.0.conv3.0", "conv", 27, 18, 3),
    ("layers.0.conv5.0", "conv", 27, 18, 5),
    ("layers.0.convpool.1", "conv", 27, 18, 3),
    ("layers.1.branch1.0", "conv", 16, 108, 1),
    ("layers.1.branch2.0", "conv", 64, 108, 1),
    ("layers.1.branch2.2", "conv", 16, 64, 3),
    ("layers.1.branch3.0", "conv", 64, 108, 1),
    ("layers.1.branch3.2", "conv", 16, 64, 5),
    ("layers.1.branch4.1", "conv", 32, 108, 1),
    ("layers.3", "linear", 64, 2400, 0),
    ("last", "linear", 2, 64, 0),
]

LINEAR_LAYERS: List[Tuple[str, str, int, int, int]] = [
    ("layers.1", "linear", 256, 540, 0),
    ("layers.3", "linear", 128, 256, 0),
    ("layers.5", "linear", 128, 128, 0),
    ("layers.7", "linear", 32, 128, 0),
    ("last", "linear", 2, 32, 0),
]


def net_layers(net: str):
    if net == "inception":
        return INCEPTION_LAYERS
    if net == "linear":
        return LINEAR_LAYERS
    raise ValueError(net)


def site_shapes(net: str) -> List[Tuple[str, Tuple[int, ...]]]:
    """Sample sites in ``named_parameters`` order: weight then bias per layer (A3: the IID
    prior covers all tensors incl. biases, bayesian.py:49-64)."""
    out = []
    for name, kind, cout, cin, k in net_layers(net):
        if kind == "conv":
            out.append((name + ".weight", (cout, cin, k)))
        else:
            out.append((name + ".weight", (cout, cin)))
        out.append((name + ".bias", (cout,)))
    return out


def n_params(net: str) -> int:
    return sum(math.prod(s) for _, s in site_shapes(net))


# --------------------------------------------------------------------------------------
# bf16 operand emulation (mirrors where the HIP bf16 path rounds)
# --------------------------------------------------------------------------------------
class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _ident(t):
    return t


def rounder(emulate_bf16: bool) -> Callable[[torch.Tensor], torch.Tensor]:
    return _RoundBf16.apply if emulate_bf16 else _ident


# --------------------------------------------------------------------------------------
# Weight samplers
# --------------------------------------------------------------------------------------
def sample_normal(mu, rho, eps):
    """[3P] tyxe.guides.AutoNormal.forward (A7): w = loc + scale*eps, scale = exp(rho)
    (constraints.positive => exp transform, cf. guides/radial.py:85-95)."""
    return mu + torch.exp(rho) * eps


def sample_radial(mu, rho, eps, r):
    """RadialNormal.rsample, bayesrul/models/guides/radial.py:31-41: ONE L2 norm over the
    whole site tensor (:38), ONE scalar distance per site per particle (:37)."""
    direction = eps / torch.norm(eps, p=2)
    return mu + (direction * r) * torch.exp(rho)


# --------------------------------------------------------------------------------------
# KL / log-densities
# --------------------------------------------------------------------------------------
def kl_normal_normal(mu, rho, mu0: float, sigma0: float):
    """torch.distributions.kl._kl_normal_normal summed over the site (A5, U1)."""
    sigma = torch.exp(rho)
    var_ratio = (sigma / sigma0) ** 2
    t1 = ((mu - mu0) / sigma0) ** 2
    return (0.5 * (var_ratio + t1 - 1.0 - torch.log(var_ratio))).sum()


_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def normal_log_prob(x, loc, scale):
    """torch.distributions.Normal.log_prob."""
    return -((x - loc) ** 2) / (2.0 * scale**2) - torch.log(scale) - _LOG_SQRT_2PI


# --------------------------------------------------------------------------------------
# Variational layers
# --------------------------------------------------------------------------------------
def _apply(kind: str, x, w, b, pad: int):
    if kind == "conv":
        return F.conv1d(x, w, b, padding=pad)
    return F.linear(x, w, b)


def layer_plain(kind, x, w, b, pad, rnd=_ident):
    """Replayed sampled weights through the unmodified layer (radial guide; every
    val/test/predict pass, bayesian.py:168-250)."""
    return _apply(kind, rnd(x), rnd(w), b, pad)


def layer_lrt(kind, x, mu_w, rho_w, mu_b, rho_b, eps_out, pad, rnd=_ident):
    """[3P] tyxe.poutine.local_reparameterization (A9, U5, U6):
    loc = fn(x, mu_W, mu_b); var = fn(x^2, sigma_W^2, sigma_b^2);
    var guarded (var + (var<0)*(|var|+1e-6), detached); out = loc + sqrt(var)*eps."""
    xq = rnd(x)
    loc = _apply(kind, xq, rnd(mu_w), mu_b, pad)
    var = _apply(kind, rnd(xq * xq), rnd(torch.exp(2.0 * rho_w)), torch.exp(2.0 * rho_b), pad)
    var = var + (var.lt(0).to(var.dtype) * (var.abs() + 1e-6)).detach()
    return loc + var.sqrt() * eps_out


def layer_flipout(kind, x, mu_w, w_sample, b_sample, s_in, s_out, pad, rnd=_ident):
    """[3P] tyxe.poutine.flipout (A10, U7): out = fn(x, mu_W, b) + fn(x*s_in, W-mu_W)*s_out,
    signs per example and per channel (broadcast over L for conv1d); the sampled bias is
    added once, outside the sign product."""
    xq = rnd(x)
    out_loc = _apply(kind, xq, rnd(mu_w), b_sample, pad)
    if kind == "conv":
        si, so = s_in.unsqueeze(-1), s_out.unsqueeze(-1)
    else:
        si, so = s_in, s_out
    pert = _apply(kind, xq * si, rnd(w_sample - mu_w), None, pad) * so
    return out_loc + pert


# --------------------------------------------------------------------------------------
# Networks (functional; `layer(name, kind, x, pad)` supplies the variational layer)
# --------------------------------------------------------------------------------------
def inception_forward(x, layer, rnd=_ident):
    """Inception.forward, nets/inception.py:211-217 (+ modules :10-132).  x: [B, 30, 18].
    Returns the net output [B, 2] after softplus + Threshold(1e-9, 1e-9)."""
    h = x.transpose(2, 1)  # [B, 18, 30]
    relu = lambda t: rnd(F.relu(t))
    pool = lambda t: F.max_pool1d(t, kernel_size=3, stride=1, padding=1)
    b1 = relu(layer("layers.0.conv1.0", "conv", h, 0))
    b2 = relu(layer("layers.0.conv3.0", "conv", h, 1))
    b3 = relu(layer("layers.0.conv5.0", "conv", h, 2))
    b4 = relu(layer("layers.0.convpool.1", "conv", pool(h), 1))
    h1 = torch.cat([b1, b2, b3, b4], 1)  # [B,108,30]
    c1 = relu(layer("layers.1.branch1.0", "conv", h1, 0))
    m2 = relu(layer("layers.1.branch2.0", "conv", h1, 0))
    c2 = relu(layer("layers.1.branch2.2", "conv", m2, 1))
    m3 = relu(layer("layers.1.branch3.0", "conv", h1, 0))
    c3 = relu(layer("layers.1.branch3.2", "conv", m3, 2))
    c4 = relu(layer("layers.1.branch4.1", "conv", pool(h1), 0))
    h2 = torch.cat([c1, c2, c3, c4], 1)  # [B,80,30]
    f = h2.flatten(1)  # [B,2400]  (index c*30 + l)
    h3 = relu(layer("layers.3", "linear", f, 0))
    z = layer("last", "linear", h3, 0)
    return head_activation(z)


def linear_net_forward(x, layer, rnd=_ident):
    """Linear.forward, nets/linear.py:65-71 (out_size=2): 540->256->128->128->32->2."""
    relu = lambda t: rnd(F.relu(t))
    h = x.flatten(1)
    h = relu(layer("layers.1", "linear", h, 0))
    h = relu(layer("layers.3", "linear", h, 0))
    h = relu(layer("layers.5", "linear", h, 0))
    h = relu(layer("layers.7", "linear", h, 0))
    z = layer("last", "linear", h, 0)
    return head_activation(z)


def head_activation(z):
    """softplus then nn.Threshold(1e-9, 1e-9) on BOTH outputs (inception.py:213-214)."""
    return F.threshold(F.softplus(z), 1e-9, 1e-9)


def net_forward(net: str, x, layer, rnd=_ident):
    return inception_forward(x, layer, rnd) if net == "inception" else linear_net_forward(x, layer, rnd)


# --------------------------------------------------------------------------------------
# Likelihood (A11)
# --------------------------------------------------------------------------------------
def hetero_gaussian_loglik(pred, y):
    """[3P] tyxe.likelihoods.HeteroskedasticGaussian(N, positive_scale=False), U3:
    loc, raw = pred.chunk(2,-1); scale = softplus(raw)  (a 2nd softplus on top of the net's);
    returns sum_b log N(y_b; loc_b, scale_b).  y: [B]."""
    loc = pred[..., 0]
    scale = F.softplus(pred[..., 1])
    return normal_log_prob(y, loc, scale).sum()


def aggregate_predictions(preds):
    """U4: precision-weighted mean; scale = sqrt(mean(s^2) + var(loc)).  preds [S,B,2]
    (net outputs); the likelihood's softplus is applied to the scale column first."""
    loc = preds[..., 0]
    scale = F.softplus(preds[..., 1])
    prec = scale.pow(-2)
    agg_loc = (loc * prec).sum(0) / prec.sum(0)
    agg_scale = (scale.pow(2).mean(0) + loc.var(0)).sqrt()
    return torch.stack([agg_loc, agg_scale], -1)


def predictive_aggregate(out):
    """BNN.predict_step / test_step aggregation, bayesian.py:212-215,241-249.
    out [S,B,2] = raw net outputs (single softplus).  Returns dict of [B] tensors."""
    loc, scale = out[:, :, 0], out[:, :, 1]
    ep_var = loc.var(0)
    al_var = (scale**2).mean(0)
    std = (al_var + ep_var).sqrt()
    pred = loc.mean(0)
    return {"ep_vars": ep_var, "al_vars": al_var, "stds": std, "preds": pred}


# --------------------------------------------------------------------------------------
# Metrics called inside every step (A17)
# --------------------------------------------------------------------------------------
def sharpness(sigma_hat):
    """results/metrics.py:210-213."""
    return torch.sqrt(torch.square(sigma_hat).mean())


def rms_calibration_error(y_pred, y_std, y_true, num_bins: int = 100):
    """results/metrics.py:216-274, prop_type='interval'; device taken from the tensors
    (the reference's ``get_device()`` = -1 on CPU raises)."""
    dev, dt = y_true.device, y_pred.dtype
    exp_p = torch.linspace(0, 1, num_bins, device=dev, dtype=dt)
    z = ((y_pred - y_true).flatten() / y_std.flatten()).reshape(-1, 1)
    nrm = torch.distributions.Normal(torch.zeros(1, device=dev, dtype=dt), torch.ones(1, device=dev, dtype=dt))
    lo = nrm.icdf(0.5 - exp_p / 2.0)
    hi = nrm.icdf(0.5 + exp_p / 2.0)
    within = (z >= lo) * (z <= hi)
    obs_p = within.sum(0).flatten() / y_pred.numel()
    return torch.sqrt(torch.mean(torch.square(exp_p - obs_p)))


def nasa_score(y_true, y_pred):
    """results/metrics.py:205-207."""
    d = y_pred - y_true
    return torch.where(d > 0, torch.exp(d / 10) - 1, torch.exp(-d / 13) - 1)


# --------------------------------------------------------------------------------------
# ELBO (A2, A4, A5, A6)
# --------------------------------------------------------------------------------------
@dataclass
class ElboConfig:
    net: str = "inception"            # "inception" | "linear"
    guide: str = "normal"             # "normal" | "radial"
    fit_context: Optional[str] = "lrt"  # "lrt" | "flipout" | None  (radial forces None, bayesian.py:83)
    dataset_size: int = 238200
    win_length: int = 30
    n_features: int = 18
    prior_loc: float = 0.0
    prior_scale: float = 1.0
    emulate_bf16: bool = False

    @property
    def mode(self) -> str:
        if self.guide == "radial":
            return "radial"
        return self.fit_context or "normal"

    @property
    def c(self) -> float:
        """poutine.scale factor 1/(N*W*F), bayesian.py:111-129."""
        return 1.0 / (self.dataset_size * self.win_length * self.n_features)


@dataclass
class ParticleNoise:
    """Noise of ONE MC particle.  Keys: site names for eps_w ('<layer>.weight|bias'),
    layer names for eps_out / s_in / s_out / r keyed by site."""
    eps_w: Dict[str, torch.Tensor] = field(default_factory=dict)   # normal / radial / flipout
    r: Dict[str, torch.Tensor] = field(default_factory=dict)       # radial: scalar per site
    eps_out: Dict[str, torch.Tensor] = field(default_factory=dict)  # lrt: output-shaped
    s_in: Dict[str, torch.Tensor] = field(default_factory=dict)    # flipout: [B, Cin]
    s_out: Dict[str, torch.Tensor] = field(default_factory=dict)   # flipout: [B, Cout]


def make_noise(cfg: ElboConfig, B: int, S: int, gen: torch.Generator, dtype=torch.float64,
               mode: Optional[str] = None) -> List[ParticleNoise]:
    """Draw all noise of a step with a seeded generator (the injected-noise protocol)."""
    mode = mode or cfg.mode
    L = cfg.win_length
    out = []
    for _ in range(S):
        pn = ParticleNoise()
        for name, shape in site_shapes(cfg.net):
            if mode in ("normal", "radial", "flipout"):
                pn.eps_w[name] = torch.randn(shape, generator=gen, dtype=dtype)
            if mode == "radial":
                pn.r[name] = torch.randn(1, generator=gen, dtype=dtype)
        for lname, kind, cout, cin, k in net_layers(cfg.net):
            if mode == "lrt":
                shp = (B, cout, L) if kind == "conv" else (B, cout)
                pn.eps_out[lname] = torch.randn(shp, generator=gen, dtype=dtype)
            if mode == "flipout":
                pn.s_in[lname] = (torch.rand((B, cin), generator=gen, dtype=dtype) > 0.5).to(dtype) * 2 - 1
                pn.s_out[lname] = (torch.rand((B, cout), generator=gen, dtype=dtype) > 0.5).to(dtype) * 2 - 1
        out.append(pn)
    return out


def _particle_layer_fn(cfg: ElboConfig, mode: str, mu, rho, pn: ParticleNoise, rnd):
    """Builds `layer(name, kind, x, pad)` for one particle and returns it together with a
    dict that collects the sampled weights (needed by Trace_ELBO)."""
    sampled: Dict[str, torch.Tensor] = {}

    def w_of(site):
        if site not in sampled:
            if mode == "radial":
                sampled[site] = sample_radial(mu[site], rho[site], pn.eps_w[site], pn.r[site])
            else:
                sampled[site] = sample_normal(mu[site], rho[site], pn.eps_w[site])
        return sampled[site]

    def layer(name, kind, x, pad):
        ws, bs = name + ".weight", name + ".bias"
        if mode == "lrt":
            return layer_lrt(kind, x, mu[ws], rho[ws], mu[bs], rho[bs], pn.eps_out[name], pad, rnd)
        if mode == "flipout":
            return layer_flipout(kind, x, mu[ws], w_of(ws), w_of(bs), pn.s_in[name], pn.s_out[name], pad, rnd)
        return layer_plain(kind, x, w_of(ws), w_of(bs), pad, rnd)

    return layer, sampled, w_of


def elbo_loss(cfg: ElboConfig, mu: Dict[str, torch.Tensor], rho: Dict[str, torch.Tensor],
              x: torch.Tensor, y: torch.Tensor, noise: List[ParticleNoise],
              mode: Optional[str] = None, obs: bool = True, scaled: bool = True):
    """The quantity ``svi.step`` returns and differentiates (A4-A6, U1, U2, U10).

    normal guide (TraceMeanField_ELBO, bayesian.py:106):
        loss = (1/S) sum_s [ c*KL(q||p) - c*(N/B)*sum_b log N(y_b; loc_sb, s_sb) ]
    radial guide (Trace_ELBO, bayesian.py:108):
        loss = (1/S) sum_s [ c*(log q(w_s) - log p(w_s)) - c*(N/B)*sum_b log N(...) ]
    ``mode`` overrides the fit context (validation uses plain sampling, bayesian.py:177).
    ``obs=False, scaled=False`` gives ``svi_no_obs.evaluate_loss`` (bayesian.py:136-139,155).
    Returns (loss, aux) with aux = {kl (unscaled, particle mean), loglik (particle mean),
    preds [S,B,2]}.
    """
    mode = mode or cfg.mode
    rnd = rounder(cfg.emulate_bf16)
    S = len(noise)
    B = x.shape[0]
    c = cfg.c if scaled else 1.0
    xin = x
    total = 0.0
    kl_acc = 0.0
    ll_acc = 0.0
    preds = []
    for pn in noise:
        layer, sampled, w_of = _particle_layer_fn(cfg, mode, mu, rho, pn, rnd)
        pred = net_forward(cfg.net, xin, layer, rnd)
        preds.append(pred)
        if cfg.guide == "radial":
            kl = 0.0
            for site, _ in site_shapes(cfg.net):
                w = w_of(site)
                logq = normal_log_prob(w, mu[site], torch.exp(rho[site])).sum()
                logp = normal_log_prob(w, torch.as_tensor(cfg.prior_loc, dtype=w.dtype),
                                       torch.as_tensor(cfg.prior_scale, dtype=w.dtype)).sum()
                kl = kl + (logq - logp)
        else:
            kl = 0.0
            for site, _ in site_shapes(cfg.net):
                kl = kl + kl_normal_normal(mu[site], rho[site], cfg.prior_loc, cfg.prior_scale)
        ll = hetero_gaussian_loglik(pred, y) if obs else torch.zeros((), dtype=x.dtype)
        total = total + (c * kl - c * (cfg.dataset_size / B) * ll) / S
        kl_acc = kl_acc + kl.detach() / S
        ll_acc = ll_acc + ll.detach() / S
    aux = {"kl": kl_acc, "loglik": ll_acc, "preds": torch.stack([p.detach() for p in preds])}
    return total, aux


# --------------------------------------------------------------------------------------
# ClippedAdam (A12, U9): [3P] pyro.optim.clipped_adam.ClippedAdam, conf/model/bnn.yaml:6-10
# --------------------------------------------------------------------------------------
@dataclass
class AdamConfig:
    lr: float = 1e-4
    beta1: float = 0.95
    beta2: float = 0.999
    eps: float = 1e-8
    clip_norm: float = 15.0
    lrd: float = 1.0
    weight_decay: float = 0.0


def clipped_adam_step(p, g, m, v, step: int, lr: float, ac: AdamConfig):
    """One ClippedAdam update of one tensor (in place on p, m, v).  `step` is the 1-based
    step count AFTER increment; `lr` the current (already decayed) learning rate.
    grad.clamp_(-clip, clip) elementwise; exp_avg, exp_avg_sq; denom = sqrt(v)+eps;
    step_size = lr*sqrt(1-b2^t)/(1-b1^t); p -= step_size*m/denom."""
    g = g.clamp(-ac.clip_norm, ac.clip_norm)
    if ac.weight_decay != 0.0:
        g = g + ac.weight_decay * p
    m.mul_(ac.beta1).add_(g, alpha=1 - ac.beta1)
    v.mul_(ac.beta2).addcmul_(g, g, value=1 - ac.beta2)
    denom = v.sqrt().add_(ac.eps)
    bc1 = 1 - ac.beta1**step
    bc2 = 1 - ac.beta2**step
    step_size = lr * math.sqrt(bc2) / bc1
    p.addcdiv_(m, denom, value=-step_size)


class SviState:
    """Unconstrained variational parameters (mu, rho = log sigma) per site + optimiser
    state; the restatement of SVI.step (A4): loss_and_grads over S sequential particles,
    optimiser step on every parameter tensor, zero grads, return float(loss)."""

    def __init__(self, cfg: ElboConfig, mu0: Dict[str, torch.Tensor], q_scale: float, adam: AdamConfig,
                 dtype=torch.float64):
        self.cfg, self.adam, self.dtype = cfg, adam, dtype
        self.mu = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in mu0.items()}
        self.rho = {k: torch.full_like(v, math.log(q_scale), dtype=dtype).requires_grad_(True)
                    for k, v in mu0.items()}
        self.m = {("mu", k): torch.zeros_like(v, dtype=dtype) for k, v in mu0.items()}
        self.m.update({("rho", k): torch.zeros_like(v, dtype=dtype) for k, v in mu0.items()})
        self.v = {k: torch.zeros_like(t) for k, t in self.m.items()}
        self.t = 0
        self.lr = adam.lr

    def loss_and_grads(self, x, y, noise, mode=None):
        for p in list(self.mu.values()) + list(self.rho.values()):
            p.grad = None
        loss, aux = elbo_loss(self.cfg, self.mu, self.rho, x.to(self.dtype), y.to(self.dtype), noise, mode)
        loss.backward()
        return loss.detach(), aux

    def step(self, x, y, noise, mode=None):
        loss, aux = self.loss_and_grads(x, y, noise, mode)
        self.t += 1
        self.lr *= self.adam.lrd  # pyro ClippedAdam decays the group lr at the top of step()
        with torch.no_grad():
            for k in self.mu:
                for kind, p in (("mu", self.mu[k]), ("rho", self.rho[k])):
                    g = p.grad if p.grad is not None else torch.zeros_like(p)
                    clipped_adam_step(p, g, self.m[(kind, k)], self.v[(kind, k)], self.t, self.lr, self.adam)
        return float(loss), aux

    def evaluate_loss(self, x, y, noise, mode="plain", obs=True, scaled=True):
        # "plain" = weights sampled from the guide and replayed through the unmodified layers
        # (validation / test / predict run outside fit_ctxt, bayesian.py:168-250).
        if mode == "plain":
            mode = "radial" if self.cfg.guide == "radial" else "normal"
        with torch.no_grad():
            m = mode
            loss, aux = elbo_loss(self.cfg, self.mu, self.rho, x.to(self.dtype), y.to(self.dtype), noise,
                                  mode=m, obs=obs, scaled=scaled)
        return float(loss), aux


# --------------------------------------------------------------------------------------
# Deterministic init helper (utils/miscellaneous.py:53-63)
# --------------------------------------------------------------------------------------
def init_mu0(net: str, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """mu0 <- freshly constructed torch layers in the reference's construction order, then
    `weights_init` (xavier_normal_ for Conv1d weights, kaiming_normal_ for Linear weights);
    biases keep torch's default init."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, kind, cout, cin, k in net_layers(net):
        if kind == "conv":
            fan_in, fan_out = cin * k, cout * k
            std = math.sqrt(2.0 / (fan_in + fan_out))
            w = torch.randn((cout, cin, k), generator=g, dtype=dtype) * std
        else:
            fan_in = cin
            std = math.sqrt(2.0 / fan_in)
            w = torch.randn((cout, cin), generator=g, dtype=dtype) * std
        bound = 1.0 / math.sqrt(fan_in)
        b = (torch.rand((cout,), generator=g, dtype=dtype) * 2 - 1) * bound
        out[name + ".weight"] = w
        out[name + ".bias"] = b
    return out
