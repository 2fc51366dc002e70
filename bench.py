#!/usr/bin/env python3
"""bench.py — ELBO-step MC-samples x windows / sec of the MI355X SVI/ELBO path.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run,
one rank per GPU, RCCL).  A step = one full svi.step over one batch of synthetic
N-CMAPSS-shaped windows: sample -> forward -> NLL -> backward -> [all-reduce] -> ClippedAdam.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# hyper-parameters: bayesrul/conf/experiment/ncmapss_{lrt,fo,rad}.yaml:17-25
WORKLOADS = {
    # BASELINE.json configs[2]: Flipout BNN (Conv1d net), 10 MC samples
    "flipout_conv_s10": dict(net="inception", guide="normal", fit_context="flipout", S=10, B=1000,
                             prior_scale=0.198768, q_scale=0.000214, lr=9.48e-4),
    # configs[3]: Radial BNN (Conv1d net), 20 MC samples, batch-DP
    "radial_conv_s20": dict(net="inception", guide="radial", fit_context=None, S=20, B=1000,
                            prior_scale=0.092516, q_scale=0.001241, lr=9.56e-4),
    # configs[1]: LRT BNN (Linear net), 1 MC sample
    "lrt_linear_s1": dict(net="linear", guide="normal", fit_context="lrt", S=1, B=1000,
                          prior_scale=0.138793, q_scale=0.001351, lr=8.57e-4),
    "lrt_conv_s1": dict(net="inception", guide="normal", fit_context="lrt", S=1, B=1000,
                        prior_scale=0.138793, q_scale=0.001351, lr=8.57e-4),
    # configs[4]: Flipout-trained Conv BNN, 100-sample predictive pass (tasks/predict.py:24-64, bayesian.py:231-250):
    # plain Normal sampling, forward only, 10,000 windows per batch (conf/datamodule/ncmapss.yaml:4); a "step" is
    # one predictive pass over the batch incl. the ep/al variance aggregation
    "predict_conv_s100": dict(net="inception", guide="normal", fit_context="flipout", S=100, B=10000,
                              prior_scale=0.198768, q_scale=0.000214, lr=9.48e-4, predict=True, chunk_particles=10),
}
N_DATA = 238200

# algorithmic MACs per sample-window of each branch group; 1 MAC = 2 FLOP
GROUP_MACS = {
    "inception": [30 * 27 * (18 + 54 + 90 + 54), 30 * 108 * (16 + 64 + 64 + 32), 30 * (192 * 16 + 320 * 16),
                  2400 * 64, 64 * 2],
    "linear": [540 * 256, 256 * 128, 128 * 128, 128 * 32, 32 * 2],
}
# algorithmic HBM bytes per MC-sample x window of each (kernel kind, branch group), Inception with
# bf16-plane storage (activation = hi + lo planes = 4 B/elem forward, hi only = 2 B/elem backward;
# gradients 2 B/elem): read + written, LRT adds a q plane
def group_bytes(net, kind, grp, em, S):
    if net != "inception":
        return None
    L = 30
    q = 2 if em == 1 else 0  # bytes/elem of the LRT q plane
    fwd = [L * 32 * 4 / S + L * 128 * (4 + q), L * 128 * 4 + L * 176 * (4 + q), L * 128 * 4 + L * 32 * (4 + q),
           L * 80 * 4 + 64 * (4 + q), 64 * 4 + 2 * 4]
    dx = [0, L * 176 * (4 + q) + 2 * L * 128 * 2, L * 32 * (4 + q) + L * 128 * 2, 64 * (4 + q) + L * 80 * 2, 2 * 8 + 64 * 2]
    dw = [L * 32 * 2 / S + L * 128 * (4 + q), L * 128 * 2 + L * 176 * (4 + q), L * 128 * 2 + L * 32 * (4 + q),
          L * 80 * 2 + 64 * (4 + q), 64 * 2 + 2 * 8]
    return {"fwd": fwd, "dx": dx, "dw": dw}[kind][grp]


PMC_SUMMARY = "r01_final_flipout_conv_s10_pmc_summary.csv"
PEAK_TFLOPS = {"bf16x3": 2500.0, "f32": 157.3}  # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


def synth(B_total, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B_total, 30, 18, generator=g)
    y = torch.randint(0, 100, (B_total,), generator=g).float()
    return x, y


def mu0_for(net):
    """Seeded init with the reference's initialisers (utils/miscellaneous.py:53-63): xavier-normal
    conv / kaiming-normal linear weights, torch-default biases."""
    import math
    from bayesrul_amd.models.nets.spec import net_layers
    g = torch.Generator().manual_seed(0)
    out = {}
    for name, conv, cout, cin, k in net_layers(net):
        if conv:
            std = math.sqrt(2.0 / (cin * k + cout * k))
            w = torch.randn(cout, cin, k, generator=g) * std
            fan_in = cin * k
        else:
            w = torch.randn(cout, cin, generator=g) * math.sqrt(2.0 / cin)
            fan_in = cin
        out[name + ".weight"] = w
        out[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) / math.sqrt(fan_in)
    return out


def cpu_baseline(wl, seconds_target=15.0):
    """CPU restatement of the reference path (oracle/, kind 'port') timed on this box's host
    cores: fp32, sequential particle loop, per-layer F.conv1d / F.linear, autograd, ClippedAdam."""
    from oracle import restatement as R
    # the 1-GPU box grants a 16-core CPU share; more threads than that only oversubscribes
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    B, S = 100, wl["S"]
    cfg = R.ElboConfig(net=wl["net"], guide=wl["guide"], fit_context=wl["fit_context"], dataset_size=N_DATA,
                       prior_scale=wl["prior_scale"])
    st = R.SviState(cfg, R.init_mu0(wl["net"], 0, torch.float32), wl["q_scale"], R.AdamConfig(lr=wl["lr"]),
                    dtype=torch.float32)
    x, y = synth(B)
    gen = torch.Generator().manual_seed(4321)
    noise = R.make_noise(cfg, B, S, gen, dtype=torch.float32)
    st.step(x, y, noise)  # warm-up
    n, t0 = 0, time.time()
    while True:
        st.step(x, y, noise)
        n += 1
        if time.time() - t0 > seconds_target or n >= 100:
            break
    dt = (time.time() - t0) / n
    return {"value": S * B / dt, "unit": "MC-samples*windows/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of B={B}, S={S}, fp32 torch CPU restatement ({dt * 1e3:.1f} ms/step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="flipout_conv_s10", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="windows per GPU (default: workload's)")
    ap.add_argument("--prec", default="bf16x3", choices=["bf16x3", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    from bayesrul_amd.engine import AdamHyper, SviEngine
    from bayesrul_amd.parallel import dp_step

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    S, B = wl["S"], wl["B"]
    predict = bool(wl.get("predict"))
    eng = SviEngine(net=wl["net"], guide=wl["guide"], fit_context=wl["fit_context"], prec=args.prec, max_particles=S,
                    max_batch=B, device=dev, max_windows=wl.get("chunk_particles", 0) * B)
    eng.init_params(mu0_for(wl["net"]), wl["q_scale"])
    xg, yg = synth(B * world)
    x = xg[rank * B:(rank + 1) * B].contiguous().to(dev)
    y = yg[rank * B:(rank + 1) * B].contiguous().to(dev)
    hyp = AdamHyper(lr=wl["lr"], betas=(0.95, 0.999), clip_norm=15.0)

    def one_step():
        if predict:   # windows shard, no collective
            return eng.predict(x, S, seed=4321, want_samples=False)[0]
        if world > 1:
            return dp_step(eng, x, y, S, N_DATA, 0.0, wl["prior_scale"], hyp, rank, world, seed=4321)
        return eng.step(x, y, S, N_DATA, 0.0, wl["prior_scale"], hyp, seed=4321)

    ncontr = 2 if (wl["fit_context"] in ("lrt", "flipout") and not predict) else 1
    em = 0 if predict else {"lrt": 1, "flipout": 2}.get(wl["fit_context"], 0)
    macs = GROUP_MACS[wl["net"]]

    def by_symbol(prof, nsteps):
        """{symbol: [ms, launches, algorithmic flops, algorithmic bytes, tags]} of the group kernels.
        One symbol may serve several branch groups; its algorithmic FLOPs per launch = (sum over its
        launches of 2*MAC*contractions*S*B) / launches, so achieved = total FLOPs / total time."""
        agg, seen = {}, set()
        for (kind, grp), (tot_ms, cnt) in prof.items():
            if kind not in ("fwd", "dx", "dw", "pool_bwd"):
                continue
            sym = eng.profile_symbol(kind, grp) or f"{kind}[{grp}]"
            a = agg.setdefault(sym, [0.0, 0, 0.0, 0.0, []])
            a[0] += tot_ms
            a[1] += cnt
            a[4].append((kind, grp))
            if kind == "pool_bwd" or (kind, grp) in seen:
                continue
            seen.add((kind, grp))
            a[2] += 2.0 * macs[grp] * ncontr * S * B * nsteps
            a[3] += (group_bytes(wl["net"], kind, grp, em, S) or 0.0) * S * B * nsteps
        return agg

    # warm-up; its last steps run with every launch recorded to find the dominant kernel symbol
    npre = min(3, args.warmup)
    for _ in range(args.warmup - npre):
        res = one_step()
    eng.profile(True)
    for _ in range(npre):
        res = one_step()
    torch.cuda.synchronize(dev)
    pre = by_symbol(eng.profile_read(), max(1, npre)) if npre else {}
    eng.profile(False)
    dom_tags = max(pre.items(), key=lambda kv: kv[1][0])[1][4] if pre else None

    # timed region: K steps; HIP events (on the launch stream, inside the library) bracket only the
    # launches of the dominant symbol -- events around every kernel cost ~10 us per launch
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    eng.profile(True, only=dom_tags)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(res[0].flatten()[0]) if predict else float(res[0])   # predict: first aggregated prediction

    # separate untimed pass: every launch recorded -> per-kernel table of the report
    npost = min(args.steps, 10)
    eng.profile(True)
    for _ in range(npost):
        one_step()
    torch.cuda.synchronize(dev)
    post = eng.profile_read()
    eng.profile(False)

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = S * B * world / (dt / args.steps)
        agg = by_symbol(prof, args.steps)
        sym, (tot_ms, cnt, flops_tot, bytes_tot, _) = max(agg.items(), key=lambda kv: kv[1][0])
        flops_launch = flops_tot / cnt
        avg_s = tot_ms / cnt * 1e-3
        achieved = flops_launch / avg_s / 1e12
        peak = PEAK_TFLOPS[args.prec]
        # HBM bytes per launch of that symbol from the committed PMC passes (profiles/, same workload):
        # (2*FETCH_SIZE + WRITE_SIZE) KB, the gfx950 read correction of MI355X_MICROARCH.md applied
        traffic = None
        pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        if args.workload == "flipout_conv_s10" and args.prec == "bf16x3" and not args.batch and os.path.exists(pmc):
            import csv
            for r in csv.DictReader(open(pmc)):
                if sym in r["kernel"]:
                    traffic = (2 * float(r["FETCH_SIZE_KB_per_launch"]) + float(r["WRITE_SIZE_KB_per_launch"])) * 1024
                    break
        bytes_launch = bytes_tot / cnt
        kernels = {f"{k[0]}[{k[1]}]": round(v[0] / npost, 4) for k, v in sorted(post.items())}
        out = {
            "metric": ("predictive-pass MC-samples x windows/sec, Conv BNN on N-CMAPSS" if predict else
                       "ELBO-step MC-samples x windows/sec, Conv BNN on N-CMAPSS"),
            "value": value, "unit": "MC-samples*windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
            "config": {"workload": args.workload, "net": wl["net"],
                       "estimator": "plain-normal predictive" if predict else (wl["fit_context"] or wl["guide"]),
                       "mc_samples": S, "windows_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "loss": loss},
            "roofline": {"bound": "mfma", "kernel": sym, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "avg_launch_us": avg_s * 1e6, "flops_per_launch": flops_launch,
                         "launches_per_step": cnt / args.steps},
            "roofline_hbm": {"bound": "hbm", "kernel": sym, "achieved": bytes_launch / avg_s / 1e9, "peak": 8000.0,
                             "unit": "GB/s", "frac": bytes_launch / avg_s / 1e9 / 8000.0,
                             "algorithmic_bytes_per_launch": bytes_launch, "traffic": traffic},
            "kernel_ms_per_step": kernels,   # separate pass with events around every launch
            "kernel_symbols": {f"{k[0]}[{k[1]}]": eng.profile_symbol(*k) for k in sorted(post)},
        }
        if world == 1 and not args.no_cpu_baseline and not predict:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
