"""Data-parallel step on world_size 2 (gloo, CPU): sharding + the single flat all-reduce of
[2P+2] (bayesrul_amd/parallel.py).  The per-rank gradient producer here is the oracle (the HIP
engine needs a GPU); what is under test is the DP arithmetic: mean over ranks of the local
gradients == gradient of the global batch, with rank-invariant weight noise and globally
indexed per-window noise."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import restatement as R

NET, B_GLOBAL, S = "linear", 6, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(mode):
    cfg = R.ElboConfig(net=NET, guide="radial" if mode == "radial" else "normal",
                       fit_context=mode if mode in ("lrt", "flipout") else None, prior_scale=0.14)
    mu0 = R.init_mu0(NET, 0, torch.float64)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B_GLOBAL, 30, 18, generator=g, dtype=torch.float64)
    y = torch.randint(0, 100, (B_GLOBAL,), generator=g).double()
    noise = R.make_noise(cfg, B_GLOBAL, S, torch.Generator().manual_seed(4321))
    return cfg, mu0, x, y, noise


def _slice_noise(noise, lo, hi):
    out = []
    for pn in noise:
        q = R.ParticleNoise(eps_w=pn.eps_w, r=pn.r)  # weight-level noise: identical on every rank
        q.eps_out = {k: v[lo:hi] for k, v in pn.eps_out.items()}  # per-window noise: global index
        q.s_in = {k: v[lo:hi] for k, v in pn.s_in.items()}
        q.s_out = {k: v[lo:hi] for k, v in pn.s_out.items()}
        out.append(q)
    return out


def _flat_grad(st, loss, kl):
    sites = [s for s, _ in R.site_shapes(NET)]
    return torch.cat([torch.cat([st.mu[s].grad.flatten() for s in sites]),
                      torch.cat([st.rho[s].grad.flatten() for s in sites]),
                      torch.tensor([float(loss), float(kl)], dtype=torch.float64)])


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bayesrul_amd.parallel import allreduce_mean_, shard_bounds
        torch.set_num_threads(1)
        cfg, mu0, x, y, noise = _setup(mode)
        lo, hi = shard_bounds(B_GLOBAL, rank, world)
        st = R.SviState(cfg, mu0, 0.02, R.AdamConfig(lr=1e-3))
        loss, aux = st.loss_and_grads(x[lo:hi], y[lo:hi], _slice_noise(noise, lo, hi))
        buf = _flat_grad(st, loss, aux["kl"])
        scale = allreduce_mean_(buf, world)
        if rank == 0:
            q.put((buf * scale).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["lrt", "flipout", "radial"])
def test_two_rank_mean_gradient_equals_global_batch(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg, mu0, x, y, noise = _setup(mode)
    st = R.SviState(cfg, mu0, 0.02, R.AdamConfig(lr=1e-3))
    loss, aux = st.loss_and_grads(x, y, noise)
    ref = _flat_grad(st, loss, aux["kl"])
    assert torch.allclose(got, ref, rtol=1e-9, atol=1e-12), float((got - ref).abs().max())


def test_shard_bounds():
    from bayesrul_amd.parallel import shard_bounds
    assert [shard_bounds(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    with pytest.raises(RuntimeError):
        shard_bounds(7, 0, 2)
