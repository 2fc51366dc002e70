#!/usr/bin/env python3
"""bench.py — ELBO-step MC-samples x windows / sec of the MI355X SVI/ELBO path.

Contract: `python bench.py --gpus N --steps K --warmup W`.  A step = one full svi.step over one batch of
synthetic N-CMAPSS-shaped windows resident in HBM: sample -> forward -> NLL -> backward -> [all-reduce] ->
ClippedAdam.  N > 1: one rank per GPU over RCCL, either under an outer `python -m torch.distributed.run ...`
(RANK / WORLD_SIZE set) or launched plainly — then this script starts that launcher itself, BEFORE any GPU call,
and exits with its code.  Rank 0 prints ONE JSON line.

Besides the contract's fields the line carries
  roofline      dominant kernel symbol: algorithmic FLOPs per launch / its mean launch duration
                (HIP events recorded inside the library on the launch stream during the timed region) vs the dense MFMA peak
  hbm           algorithmic bytes per step next to the measured PMC traffic of the committed
                rocprofv3 passes (profiles/): a traffic ratio, not a roofline fraction
  median        median per-step time over >= 200 separately timed steps (events between steps)
  fast_plan     the same workload on the split-bf16 plan (`--prec bf16x3`: narrower than the reference's fp32, reported
                beside the judged fp32 line, never as `value`)
  b100          the same workload at the reference's batch of 100 windows (launch-bound regime)
  shipped_lrt_s1_b100   the reference's shipped LRT experiment at its own settings (1 MC sample, batch 100), exact fp32
  cpu_baseline  the CPU restatement (oracle/, kind "port") on this box's host cores: all cores at the workload's own
                batch, one thread on a smaller sample
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

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
    # (BASELINE.json names bf16 for this config: its default plan is the split-bf16 one)
    "lrt_linear_s1": dict(net="linear", guide="normal", fit_context="lrt", S=1, B=1000,
                          prior_scale=0.138793, q_scale=0.001351, lr=8.57e-4, prec="bf16x3"),
    # the reference's shipped LRT experiment (ncmapss_lrt.yaml)
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
N_PARAMS = {"inception": 187142, "linear": 192098}
# branch groups one launch of a fused kernel covers (the library reports the symbol per (kind, group) tag)
FUSED_GROUPS = {"trunk_fwd_kernel": (0, 1, 2), "trunk_dx_kernel": (1, 2), "tf_fwd_kernel": (0, 1, 2), "tf_dx_kernel": (1, 2),
                "tf_dx_lrt_kernel": (1, 2)}
# the fp32 plan's dW launches: kind 0 = block 1 + the k3 / k5 level, kind 1 = the 1x1 level (keyed by the profile tag's group)
FUSED_DW_GROUPS = {"tf_dw_kernel": {0: (0, 2), 1: (1,)}}
PEAK_TFLOPS = {"bf16x3": 2500.0, "f32": 157.3}  # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def algorithmic_bytes_per_step(net, S, B, predict):
    """windows 2,164 B each (read once per step), (mu, rho) read + written, Adam m / v read + written,
    gradient written + read = 8 * 2P * 4 B; weight noise is not algorithmic (in-kernel Philox); activations are assumed
    on chip."""
    P = N_PARAMS[net]
    if predict:
        return B * 2160 + 2 * P * 4 + 4 * B * 4
    return B * 2164 + 8 * 2 * P * 4


def synth(B_total, seed=1234):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B_total, 30, 18, generator=g)
    y = torch.randint(0, 100, (B_total,), generator=g).float()
    return x, y


def mu0_for(net):
    """Seeded init with the reference's initialisers (utils/miscellaneous.py:53-63): xavier-normal
    conv / kaiming-normal linear weights, torch-default biases."""
    import math

    import torch
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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_time(wl, B, threads, seconds_target, max_steps):
    import torch
    from oracle import restatement as R
    torch.set_num_threads(threads)
    S = wl["S"]
    cfg = R.ElboConfig(net=wl["net"], guide=wl["guide"], fit_context=wl["fit_context"], dataset_size=N_DATA,
                       prior_scale=wl["prior_scale"])
    st = R.SviState(cfg, R.init_mu0(wl["net"], 0, torch.float32), wl["q_scale"], R.AdamConfig(lr=wl["lr"]),
                    dtype=torch.float32)
    x, y = synth(B)
    noise = R.make_noise(cfg, B, S, torch.Generator().manual_seed(4321), dtype=torch.float32)
    st.step(x, y, noise)  # warm-up
    ts = []
    t_all = time.time()
    while len(ts) < max_steps and (time.time() - t_all < seconds_target or len(ts) < 2):
        t0 = time.time()
        st.step(x, y, noise)
        ts.append(time.time() - t0)
    dt = statistics.median(ts)
    return S * B / dt, dt, len(ts)


def cpu_baseline(wl):
    """CPU restatement of the reference path (oracle/, kind 'port') timed on this box's host cores: fp32, sequential
    particle loop, per-layer F.conv1d / F.linear, autograd, ClippedAdam — the structure of the reference's Pyro/TyXe
    step.  A reported baseline, not the optimisation target."""
    import torch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))   # the 1-GPU box grants a 16-core share; more threads only oversubscribe
    B = wl["B"]
    v_all, dt_all, n_all = _cpu_time(wl, B, cores, 14.0, 50)
    b1 = min(B, 100)
    v_one, dt_one, n_one = _cpu_time(wl, b1, 1, 8.0, 20)
    torch.set_num_threads(cores)
    return {"value": v_all, "unit": "MC-samples*windows/s", "cores": cores, "kind": "port",
            "sample": f"median of {n_all} steps of B={B}, S={wl['S']}, fp32 torch CPU restatement, {cores} threads "
                      f"({dt_all * 1e3:.0f} ms/step)",
            "one_thread": {"value": v_one, "sample": f"median of {n_one} steps of B={b1}, S={wl['S']} ({dt_one * 1e3:.0f} ms/step)"},
            "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "torch": torch.__version__}


def spawn_ranks(args):
    """`python bench.py --gpus N` launched plainly: start one rank per GPU through torch's launcher before any GPU
    call of this process and hand its exit code on."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="flipout_conv_s10", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="windows per GPU (default: workload's)")
    ap.add_argument("--prec", default=None, choices=["bf16x3", "f32"],
                    help="f32 = exact-fp32 MFMA, the reference's arithmetic precision (default; the judged line); bf16x3 = "
                         "split-bf16 fast plan (default of the LRT workloads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true", help="skip the fp32_plan / b100 / median companions")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    from bayesrul_amd.engine import AdamHyper, SviEngine
    from bayesrul_amd.parallel import dp_step

    wl = dict(WORKLOADS[args.workload])
    if args.prec is None:
        args.prec = wl.get("prec", "f32")
    if args.batch:
        wl["B"] = args.batch
    S, B = wl["S"], wl["B"]
    predict = bool(wl.get("predict"))
    hyp = AdamHyper(lr=wl["lr"], betas=(0.95, 0.999), clip_norm=15.0)
    xg, yg = synth(max(B, 100) * world)

    def make(prec, batch):
        eng = SviEngine(net=wl["net"], guide=wl["guide"], fit_context=wl["fit_context"], prec=prec, max_particles=S,
                        max_batch=batch, device=dev, max_windows=wl.get("chunk_particles", 0) * batch)
        eng.init_params(mu0_for(wl["net"]), wl["q_scale"])
        x = xg[rank * batch:(rank + 1) * batch].contiguous().to(dev)
        y = yg[rank * batch:(rank + 1) * batch].contiguous().to(dev)

        def one_step():
            if predict:   # windows shard, no collective
                return eng.predict(x, S, seed=4321, want_samples=False)[0]
            if world > 1:
                return dp_step(eng, x, y, S, N_DATA, 0.0, wl["prior_scale"], hyp, rank, world, seed=4321)
            return eng.step(x, y, S, N_DATA, 0.0, wl["prior_scale"], hyp, seed=4321, keep=False)   # result read at once: no copy kernel
        return eng, one_step

    eng, one_step = make(args.prec, B)
    ncontr = 2 if (wl["fit_context"] in ("lrt", "flipout") and not predict) else 1
    macs = GROUP_MACS[wl["net"]]

    def by_symbol(prof, nsteps):
        """{symbol: [ms, launches, algorithmic flops, tags]} of the contraction kernels.  One symbol may serve several
        branch groups (and a fused kernel covers several per launch); its algorithmic FLOPs per launch = (sum over its
        launches of 2*MAC*contractions*S*B) / launches, so achieved = total FLOPs / total time."""
        agg, seen = {}, set()
        for (kind, grp), (tot_ms, cnt) in prof.items():
            if kind not in ("fwd", "dx", "dw", "pool_bwd"):
                continue
            sym = eng.profile_symbol(kind, grp) or f"{kind}[{grp}]"
            a = agg.setdefault(sym, [0.0, 0, 0.0, []])
            a[0] += tot_ms
            a[1] += cnt
            a[3].append((kind, grp))
            if kind == "pool_bwd" or (kind, grp) in seen:
                continue
            seen.add((kind, grp))
            base = sym.split("<")[0]
            groups = FUSED_DW_GROUPS[base][grp] if base in FUSED_DW_GROUPS else FUSED_GROUPS.get(base, (grp,))
            if kind == "dx":   # block 1 needs no dX (its input is the data)
                groups = [g for g in groups if g > 0]
            a[2] += sum(2.0 * macs[g] * ncontr * S * B * nsteps for g in groups)
        return agg

    def timed_region(step_fn, engine, nsteps, only):
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        if only is not None:
            engine.profile(True, only=only)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            res = step_fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, res

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
    dom_tags = max(pre.items(), key=lambda kv: kv[1][0])[1][3] if pre else None

    # timed region: EXACTLY K steps; HIP events (on the launch stream, inside the library) bracket only the launches
    # of the dominant symbol -- events around every kernel cost ~5 us per launch
    dt, res = timed_region(one_step, eng, args.steps, dom_tags)
    prof = eng.profile_read()
    eng.profile(False)
    loss = float(res[0].flatten()[0]) if predict else float(res[0])   # predict: first aggregated prediction

    # separate untimed pass: every launch recorded -> per-kernel table of the report
    npost = min(args.steps, 10)
    eng.profile(True)
    for _ in range(npost):
        one_step()
    torch.cuda.synchronize(dev)
    post = eng.profile_read()
    eng.profile(False)
    agg = by_symbol(prof, args.steps)
    kernel_symbols = {f"{k[0]}[{k[1]}]": eng.profile_symbol(*k) for k in sorted(post)}

    def median_ms(step_fn, n):
        """median per-step time: events between the steps on the launch stream, one synchronisation at the end"""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            step_fn()
            evs[i + 1].record()
        torch.cuda.synchronize(dev)
        return statistics.median(evs[i].elapsed_time(evs[i + 1]) for i in range(n))

    companions = {}
    if not args.no_companions and world == 1:
        nmed = 200 if not predict else 20
        companions["median"] = {"timed_steps": nmed, "ms_per_step": median_ms(one_step, nmed)}
        companions["median"]["value"] = S * B / (companions["median"]["ms_per_step"] * 1e-3)
        if not predict and not (wl["net"] == "inception" and wl["fit_context"] == "lrt"):   # (LRT / Inception: fp32 plan only)
            # the other precision plan of the same workload: the judged line is the exact-fp32 plan (the reference trains in
            # fp32); the split-bf16 plan is the fast, narrower alternative
            other = "bf16x3" if args.prec == "f32" else "f32"
            engo, stepo = make(other, B)
            for _ in range(3):
                stepo()
            m = median_ms(stepo, 50)
            companions["fast_plan" if other == "bf16x3" else "fp32_plan"] = {
                "timed_steps": 50, "ms_per_step": m, "value": S * B / (m * 1e-3),
                "dtype": ("bf16x3: forward mean path split-bf16 (3 MFMAs), second contraction and backward single bf16, fp32 "
                          "accumulate - narrower than the reference's fp32" if other == "bf16x3"
                          else "f32 (v_mfma_f32_16x16x4_f32)")}
            del engo
            torch.cuda.empty_cache()
            # the reference's batch size (conf/datamodule/ncmapss.yaml: batch_size 100): launch-bound regime
            if B != 100:
                eng100, step100 = make(args.prec, 100)
                for _ in range(5):
                    step100()
                m = median_ms(step100, 200)
                companions["b100"] = {"timed_steps": 200, "windows_per_gpu": 100, "ms_per_step": m,
                                      "value": S * 100 / (m * 1e-3)}
                del eng100
                torch.cuda.empty_cache()
            # the reference's shipped LRT experiment at its own settings (conf/experiment/ncmapss_lrt.yaml:17-28: LRT, 1 MC sample;
            # conf/datamodule/ncmapss.yaml:3: batch 100): every kernel of the step is at its fixed cost there
            if args.workload == "flipout_conv_s10":
                w2 = WORKLOADS["lrt_conv_s1"]
                e2 = SviEngine(net=w2["net"], guide=w2["guide"], fit_context=w2["fit_context"], prec="f32", max_particles=1, max_batch=100,
                               device=dev)
                e2.init_params(mu0_for(w2["net"]), w2["q_scale"])
                x2, y2 = xg[:100].contiguous().to(dev), yg[:100].contiguous().to(dev)
                h2 = AdamHyper(lr=w2["lr"], betas=(0.95, 0.999), clip_norm=15.0)

                def step2():
                    return e2.step(x2, y2, 1, N_DATA, 0.0, w2["prior_scale"], h2, seed=4321, keep=False)
                for _ in range(5):
                    step2()
                m = median_ms(step2, 200)
                companions["shipped_lrt_s1_b100"] = {"timed_steps": 200, "workload": "lrt_conv_s1", "mc_samples": 1, "windows_per_gpu": 100,
                                                     "dtype": "f32", "ms_per_step": m, "value": 100 / (m * 1e-3)}
                del e2
                torch.cuda.empty_cache()

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = S * B * world / (dt / args.steps)
        sym, (tot_ms, cnt, flops_tot, _) = max(agg.items(), key=lambda kv: kv[1][0])
        flops_launch = flops_tot / cnt
        avg_s = tot_ms / cnt * 1e-3
        achieved = flops_launch / avg_s / 1e12
        peak = PEAK_TFLOPS[args.prec]
        # HBM bytes from the committed PMC passes (profiles/, same workload): (2*FETCH_SIZE + WRITE_SIZE) KB per
        # launch, the gfx950 read correction of MI355X_MICROARCH.md applied
        traffic, step_traffic = None, None
        pmc = os.path.join(ROOT, "profiles", f"r03_{args.workload}_{args.prec}_pmc_summary.csv")
        if not args.batch and os.path.exists(pmc):
            import csv
            step_traffic = 0.0
            for r in csv.DictReader(open(pmc)):
                b = (2 * float(r["FETCH_SIZE_KB_per_launch"]) + float(r["WRITE_SIZE_KB_per_launch"])) * 1024
                lps = float(r.get("launches_per_step", 1) or 1)
                if "fillBuffer" in r["kernel"] and lps < 1.0:
                    continue   # the one-off zero fill of the workspace at bind time, not step traffic
                step_traffic += b * lps
                if sym in r["kernel"]:
                    traffic = b
        alg_bytes = algorithmic_bytes_per_step(wl["net"], S, B, predict)
        kernels = {f"{k[0]}[{k[1]}]": round(v[0] / npost, 4) for k, v in sorted(post.items())}
        total_flops = sum(2.0 * m for m in macs) * (1 if predict else 3) * ncontr * S * B
        out = {
            "metric": ("predictive-pass MC-samples x windows/sec, Conv BNN on N-CMAPSS" if predict else
                       "ELBO-step MC-samples x windows/sec, Conv BNN on N-CMAPSS"),
            "value": value, "unit": "MC-samples*windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
            "config": {"workload": args.workload, "net": wl["net"],
                       "estimator": "plain-normal predictive" if predict else (wl["fit_context"] or wl["guide"]),
                       "mc_samples": S, "windows_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "loss": loss,
                       "arithmetic": ("forward mean path split-bf16 (hi+lo, 3 MFMAs, fp32 accumulate); second "
                                      "contraction and backward contractions single bf16" if args.prec == "bf16x3"
                                      else "exact fp32 MFMA")},
            "roofline": {"bound": "mfma", "kernel": sym, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "avg_launch_us": avg_s * 1e6, "flops_per_launch": flops_launch,
                         "launches_per_step": cnt / args.steps,
                         "whole_step": {"algorithmic_flops": total_flops, "achieved": total_flops / (ms * 1e-3) / 1e12,
                                        "frac": total_flops / (ms * 1e-3) / 1e12 / peak}},
            "hbm": {"algorithmic_bytes_per_step": alg_bytes, "measured_bytes_per_step": step_traffic,
                    "traffic_ratio": (step_traffic / alg_bytes) if step_traffic else None,
                    "algorithmic_GBps": alg_bytes / (ms * 1e-3) / 1e9, "peak_GBps": HBM_PEAK_GBS},
            "kernel_ms_per_step": kernels,   # separate pass with events around every launch
            "kernel_symbols": kernel_symbols,
        }
        out.update(companions)
        if world == 1 and not args.no_cpu_baseline and not predict:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
