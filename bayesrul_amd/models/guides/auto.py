"""Guide objects with the surface of TyXe's `AutoNormal` and the reference's `AutoRadial`
(bayesrul/models/guides/radial.py:44-144): `get_loc / get_scale / get_detached_distributions /
forward`.  They are *views* on the engine's flat (mu, log sigma) buffers; the sampling
arithmetic itself runs in the HIP kernels (csrc/kernels_misc.h: prep_weights_kernel)."""
from __future__ import annotations

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


class _AutoGuide:
    dist_cls = td.Normal

    def __init__(self, engine, init_scale: float = 1e-1, train_loc: bool = True, train_scale: bool = True,
                 max_guide_scale: Optional[float] = None):
        if max_guide_scale is not None:
            raise RuntimeError("max_guide_scale (interval constraint) is not used by any reference config")
        if not (train_loc and train_scale):
            raise RuntimeError("train_loc / train_scale = False are not used by any reference config")
        self.engine = engine
        self.init_scale = init_scale

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
