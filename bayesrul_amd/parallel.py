"""Batch data-parallel SVI step: one process per GPU, windows sharded across ranks, ONE
all-reduce of the flat gradient buffer [2P+2] per step (RCCL over xGMI through
torch.distributed backend "nccl"; "gloo" in the CPU tests).  New functionality of the build:
the reference has no distributed path.

Rank r holds windows [r*B/G, (r+1)*B/G) of the global batch.  Each rank's loss is
c*KL - c*(N/B_r)*sum_{b in r} loglik_b, so the MEAN over ranks of the gradients equals the
single-GPU gradient on the global batch (the KL gradient is replicated).  Weight-level noise
is identical on every rank (same Philox key/step); per-window noise is indexed by the global
window id, so results do not depend on G up to reduction order.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank's windows; the global batch must divide evenly so that every rank
    scales its log-likelihood by the same N/B_r."""
    if global_batch % world != 0:
        raise RuntimeError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def allreduce_mean_(buf: torch.Tensor, world: int) -> float:
    """Sum all-reduce of the flat [2P+2] buffer (grads + loss + kl); returns the factor the
    optimiser must apply (1/world) instead of touching the buffer again."""
    if world > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return 1.0 / world


def dp_step(engine, x_local, y_local, particles, dataset_size, prior_loc, prior_scale, adam, rank: int, world: int,
            seed: int = 0, step=None):
    """svi.step on `world` GPUs: local gradient -> all-reduce -> ClippedAdam on every replica.
    Returns the [loss, kl, loglik] device tensor of the GLOBAL batch (loss, kl) / local (loglik)."""
    B = x_local.shape[0]
    res = engine.step(x_local, y_local, particles, dataset_size, prior_loc, prior_scale, None, seed=seed, step=step,
                      global_batch=B * world, global_batch_offset=rank * B, keep=False)   # cloned below
    scale = allreduce_mean_(engine.grad, world)
    engine.apply_adam(adam, grad_scale=scale)
    out = res.clone()
    out[0] = engine.grad[2 * engine.P] * scale
    out[1] = engine.grad[2 * engine.P + 1] * scale
    return out
