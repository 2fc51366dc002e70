"""Test helpers: oracle noise <-> the injectable layouts of the C ABI (BnnNoise)."""
from __future__ import annotations

from typing import Dict, List

import torch

from oracle import restatement as R


def oracle_cfg(net, mode, prior_scale=0.14, emulate_bf16=False, dataset_size=238200):
    return R.ElboConfig(net=net, guide="radial" if mode == "radial" else "normal",
                        fit_context=mode if mode in ("lrt", "flipout") else None, dataset_size=dataset_size,
                        prior_scale=prior_scale, emulate_bf16=emulate_bf16)


def to_injected(engine, cfg: R.ElboConfig, noise: List[R.ParticleNoise], B: int, mode: str = None):
    from bayesrul_amd.engine import InjectedNoise
    dev = engine.device
    mode = mode or cfg.mode
    S = len(noise)
    sites = [s for s, _ in R.site_shapes(cfg.net)]
    out = InjectedNoise()
    f = lambda t: t.to(torch.float32).contiguous().to(dev)
    if mode in ("normal", "radial", "flipout"):
        out.eps_w = f(torch.stack([torch.cat([pn.eps_w[s].flatten() for s in sites]) for pn in noise]))
    if mode == "radial":
        out.radial_r = f(torch.stack([torch.cat([pn.r[s].flatten() for s in sites]) for pn in noise]))
    layers = R.net_layers(cfg.net)
    if mode == "lrt":
        lst = []
        for lname, kind, cout, cin, k in layers:
            e = torch.stack([pn.eps_out[lname] for pn in noise])  # [S,B,Cout,L] or [S,B,Cout]
            if kind == "conv":
                e = e.permute(0, 1, 3, 2)
            lst.append(f(e))
        out.lrt_eps = lst
    if mode == "flipout":
        si, so = [], []
        for li, (lname, kind, cout, cin, k) in enumerate(layers):
            cin_img = engine.layers[li][1]
            idx = engine.cin_image_index(li)
            s = torch.stack([pn.s_in[lname] for pn in noise])  # [S,B,cin]
            img = torch.ones(S, B, cin_img, dtype=s.dtype)
            img[:, :, idx] = s
            si.append(f(img))
            so.append(f(torch.stack([pn.s_out[lname] for pn in noise])))
        out.sign_in, out.sign_out = si, so
    return out


def from_injected(engine, cfg: R.ElboConfig, inj, B: int, S: int, mode: str = None, dtype=torch.float64):
    """Inverse of to_injected (used with engine.export_noise)."""
    mode = mode or cfg.mode
    out = []
    shapes = R.site_shapes(cfg.net)
    layers = R.net_layers(cfg.net)
    for s in range(S):
        pn = R.ParticleNoise()
        off = 0
        for si, (name, shp) in enumerate(shapes):
            n = 1
            for d in shp:
                n *= d
            if mode in ("normal", "radial", "flipout"):
                pn.eps_w[name] = inj.eps_w[s, off:off + n].view(shp).to("cpu", dtype)
            if mode == "radial":
                pn.r[name] = inj.radial_r[s, si:si + 1].to("cpu", dtype)
            off += n
        for li, (lname, kind, cout, cin, k) in enumerate(layers):
            if mode == "lrt":
                e = inj.lrt_eps[li][s].to("cpu", dtype)
                pn.eps_out[lname] = e.permute(0, 2, 1).contiguous() if kind == "conv" else e
            if mode == "flipout":
                idx = engine.cin_image_index(li)
                pn.s_in[lname] = inj.sign_in[li][s].to("cpu", dtype)[:, idx]
                pn.s_out[lname] = inj.sign_out[li][s].to("cpu", dtype)
        out.append(pn)
    return out


def flat(d: Dict[str, torch.Tensor], net: str):
    return torch.cat([d[s].detach().flatten() for s, _ in R.site_shapes(net)])


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-300))


def synth_batch(B, seed=1234, dtype=torch.float32):
    """Synthetic N-CMAPSS-shaped windows: x ~ N(0,1) [B,30,18], y ~ U{0..99}."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 30, 18, generator=g, dtype=dtype)
    y = torch.randint(0, 100, (B,), generator=g).to(dtype)
    return x, y


def record(name: str, **vals):
    """Measured errors of a tolerance test: printed (pytest -s / -rP shows them) and appended to
    gpurun_out/measured_errors.jsonl, from which the bounds in the tests are set (<= 2x the measured value)."""
    import json
    import os
    row = {"test": name, **{k: (float(v) if not isinstance(v, (str, list, dict)) else v) for k, v in vals.items()}}
    print("MEASURED", json.dumps(row))
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "measured_errors.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass
