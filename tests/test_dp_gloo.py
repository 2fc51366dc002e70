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


class _OracleEngine:
    """Engine-shaped wrapper of the oracle (the `step` / `grad` / `apply_adam` / `P` surface dp_step drives):
    lets the real `bayesrul_amd.parallel.dp_step` run on CPU ranks."""

    def __init__(self, mode, noise):
        cfg, mu0, _, _, _ = _setup(mode)
        self.st = R.SviState(cfg, mu0, 0.02, R.AdamConfig(lr=1e-3))
        self.noise = noise
        self.P = R.n_params(NET)
        self.grad = torch.zeros(2 * self.P + 2, dtype=torch.float64)
        self.steps_seen = []

    def step(self, x, y, particles, dataset_size, prior_loc, prior_scale, adam, seed=0, step=None, global_batch=0,
             global_batch_offset=0, keep=True):
        assert adam is None, "dp_step must leave the update to apply_adam (after the all-reduce)"
        lo = global_batch_offset
        loss, aux = self.st.loss_and_grads(x, y, _slice_noise(self.noise, lo, lo + x.shape[0]))
        self.grad.copy_(_flat_grad(self.st, loss, aux["kl"]))
        self.steps_seen.append((global_batch, global_batch_offset))
        return torch.tensor([float(loss), float(aux["kl"]), float(aux["loglik"])], dtype=torch.float64)

    def apply_adam(self, adam, grad_scale=1.0):
        st, sites, off = self.st, [s for s, _ in R.site_shapes(NET)], 0
        st.t += 1
        with torch.no_grad():
            for kind, params in (("mu", st.mu), ("rho", st.rho)):
                for s in sites:
                    n = params[s].numel()
                    g = (self.grad[off:off + n] * grad_scale).view_as(params[s])
                    R.clipped_adam_step(params[s], g, st.m[(kind, s)], st.v[(kind, s)], st.t, st.lr, st.adam)
                    off += n


def _dp_worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bayesrul_amd.parallel import dp_step, shard_bounds
        torch.set_num_threads(1)
        cfg, mu0, x, y, noise = _setup(mode)
        lo, hi = shard_bounds(B_GLOBAL, rank, world)
        eng = _OracleEngine(mode, noise)
        outs = []
        for _ in range(2):   # two optimiser steps: the replicas must stay in lock-step
            outs.append(dp_step(eng, x[lo:hi], y[lo:hi], S, cfg.dataset_size, 0.0, cfg.prior_scale, adam=object(),
                                rank=rank, world=world))
        assert eng.steps_seen == [(B_GLOBAL, lo)] * 2
        sites = [s for s, _ in R.site_shapes(NET)]
        flat = torch.cat([eng.st.mu[s].detach().flatten() for s in sites] + [eng.st.rho[s].detach().flatten() for s in sites])
        q.put((rank, flat.numpy(), torch.stack(outs).numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["flipout", "radial"])
def test_dp_step_two_ranks_equals_single_process(mode):
    """bayesrul_amd.parallel.dp_step itself (local step with adam=None -> all_reduce(engine.grad) -> apply_adam with
    grad_scale = 1/world -> loss / kl read back from the reduced tail) on two gloo ranks == two oracle steps on the
    global batch; both replicas end with identical parameters."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(2):
        r, flat, outs = q.get(timeout=180)
        res[r] = (torch.from_numpy(flat), torch.from_numpy(outs))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg, mu0, x, y, noise = _setup(mode)
    st = R.SviState(cfg, mu0, 0.02, R.AdamConfig(lr=1e-3))
    losses = [st.step(x, y, noise) for _ in range(2)]
    sites = [s for s, _ in R.site_shapes(NET)]
    ref = torch.cat([st.mu[s].detach().flatten() for s in sites] + [st.rho[s].detach().flatten() for s in sites])
    assert torch.equal(res[0][0], res[1][0])                      # replicas in lock-step
    assert torch.allclose(res[0][0], ref, rtol=1e-9, atol=1e-12)
    for k, (lo, aux) in enumerate(losses):                        # global loss / kl come out of the reduced buffer
        assert abs(float(res[0][1][k, 0]) - lo) <= 1e-9 * abs(lo)
        assert abs(float(res[0][1][k, 1]) - float(aux["kl"])) <= 1e-9 * abs(float(aux["kl"]))


def test_shard_bounds():
    from bayesrul_amd.parallel import shard_bounds
    assert [shard_bounds(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    with pytest.raises(RuntimeError):
        shard_bounds(7, 0, 2)
