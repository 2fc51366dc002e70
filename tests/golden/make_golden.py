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


if __name__ == "__main__":
    main()
