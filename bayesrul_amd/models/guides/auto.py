"""Guide objects with the surface of TyXe's `AutoNormal` and the reference's `AutoRadial`
(bayesrul/models/guides/radial.py:44-144): `get_loc / get_scale / get_detached_distributions /
forward`.  They are *views* on the engine's flat (mu, log sigma) buffers; the sampling
arithmetic itself runs in the HIP kernels (csrc/kernels_misc.h: prep_weights_kernel)."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.distributions as td


class RadialNormal(td.Normal):
    """Normal whose rsample draws the radial perturbation mu + sigma * (eps/||eps||) * r
    (guides/radial.py:31-41); log_prob stays Normal's (this is what Trace_ELBO evaluates)."""

    def rsample(self, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        eps = torch.randn(shape, dtype=self.loc.dtype, device=self.loc.device)
        distance = torch.randn(1, device=self.loc.device)
        direction = eps / torch.norm(eps, p=2)
        return self.loc + direction * distance * self.scale


def _prior_std(method: str, weight: torch.Tensor) -> float:
    """Fan-based prior-relative scale for a string `init_scale` (guides/radial.py:66-72 calls
    `tyxe.util.calculate_prior_std` [3P]).  PARITY UNPINNED: TyXe is absent and no shipped config or fixture uses a string
    init_scale; this uses torch's fan convention (fan_out includes the receptive field, a bias has fan_in = numel), which
    may differ from TyXe's `fan_in_fan_out` for biases and conv weights.  Do not rely on it matching the reference."""
    fan_in = weight[0].numel() if weight.dim() > 1 else weight.numel()
    fan_out = weight.shape[0] * (weight[0, 0].numel() if weight.dim() > 2 else 1) if weight.dim() > 1 else weight.numel()
    if method == "radford":
        return fan_in ** -0.5
    if method == "xavier":
        return math.sqrt(2.0 / (fan_in + fan_out))
    if method == "kaiming":
        return math.sqrt(1.0 / fan_in)
    raise RuntimeError(f"unknown init_scale rule {method!r}")


class _AutoGuide:
    dist_cls = td.Normal

    def __init__(self, engine, init_scale=1e-1, train_loc: bool = True, train_scale: bool = True,
                 max_guide_scale: Optional[float] = None):
        """guides/radial.py:44-95.  `init_scale`: a number, a dict {site: tensor} or the name of a fan-based rule
        ("radford" | "xavier" | "kaiming", `tyxe.util.calculate_prior_std` [3P, from memory]); it sets rho = log(scale).
        `train_loc / train_scale = False` freeze the parameter in the fused optimiser.  `max_guide_scale` (an interval
        constraint on the scale, i.e. another unconstrained parametrisation) is not implemented by the kernels."""
        if max_guide_scale is not None:
            raise RuntimeError("max_guide_scale (interval-constrained scale) is not implemented by the MI355X kernels; "
                               "no reference config uses it")
        self.engine = engine
        self.init_scale, self.train_loc, self.train_scale = init_scale, bool(train_loc), bool(train_scale)
        with torch.no_grad():
            for name, _, _ in engine.sites:
                rho = engine.log_scale(name)
                if isinstance(init_scale, dict):
                    rho.copy_(torch.as_tensor(init_scale[name]).to(rho.device, rho.dtype).log().expand_as(rho))
                elif isinstance(init_scale, str):
                    rho.fill_(math.log(_prior_std(init_scale, engine.loc(name))))
                else:
                    rho.fill_(math.log(float(init_scale)))

    def site_names(self):
        return [s for s, _, _ in self.engine.sites]

    def get_loc(self, site_name: str) -> torch.Tensor:
        return self.engine.loc(site_name)

    def get_scale(self, site_name: str) -> torch.Tensor:
        return self.engine.log_scale(site_name).exp()

    def get_detached_distributions(self, site_names=None) -> Dict[str, td.Distribution]:
        names = self.site_names() if site_names is None else site_names
        out = {}
        for n in self.site_names():
            if n not in names:
                continue
            loc, scale = self.get_loc(n).detach().clone(), self.get_scale(n).detach().clone()
            out[n] = td.Independent(self.dist_cls(loc, scale), loc.dim())
        return out

    def forward(self, *args, **kwargs) -> Dict[str, torch.Tensor]:
        """One draw of every site (host-side convenience; the step kernels draw their own)."""
        return {n: d.rsample() for n, d in self.get_detached_distributions().items()}

    __call__ = forward


class AutoNormal(_AutoGuide):
    dist_cls = td.Normal


class AutoRadial(_AutoGuide):
    dist_cls = RadialNormal
