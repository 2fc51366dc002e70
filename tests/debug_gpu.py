"""Diagnostic (not a test): per-layer / per-site comparison of the HIP path with the oracle.
Usage on the GPU box:  python tests/debug_gpu.py [modes...] > gpurun_out/debug.log"""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import restatement as R
from tests.noise_util import oracle_cfg, rel_l2, synth_batch, to_injected
from tests.test_gpu_parity import HYP, N_DATA, _engine
from bayesrul_amd import _native as N


def capture(cfg, st, x, pn, mode):
    mu = {k: v.detach() for k, v in st.mu.items()}
    rho = {k: v.detach() for k, v in st.rho.items()}
    layer, sampled, w_of = R._particle_layer_fn(cfg, mode, mu, rho, pn, R._ident)
    cap = {}

    def layer2(name, kind, h, pad):
        o = layer(name, kind, h, pad)
        cap[name] = o
        return o

    pred = R.net_forward(cfg.net, x.double(), layer2)
    return cap, pred


def run(net, mode, prec, S=2, B=3, q_boost=20.0):
    ps, qs, lr = HYP[mode]
    qs *= q_boost
    eng = _engine(net, mode, prec, S, B)
    mu0 = R.init_mu0(net, 0, torch.float64)
    eng.init_params(mu0, qs)
    cfg = oracle_cfg(net, mode, ps)
    st = R.SviState(cfg, mu0, qs, R.AdamConfig(lr=lr))
    x, y = synth_batch(B)
    noise = R.make_noise(cfg, B, S, torch.Generator().manual_seed(4321))
    inj = to_injected(eng, cfg, noise, B)
    res, preds = eng.step(x.cuda(), y.cuda(), S, N_DATA, 0.0, ps, None, noise=inj, want_preds=True)
    torch.cuda.synchronize()
    loss_o, aux = st.loss_and_grads(x, y, noise)
    print(f"== {net} {mode} {prec}: loss dev {float(res[0]):.9g} oracle {float(loss_o):.9g} | kl {float(res[1]):.9g} / {float(aux['kl']):.9g} | ll {float(res[2]):.9g} / {float(aux['loglik']):.9g}")
    print("   preds relerr", rel_l2(preds, aux["preds"]))
    L = 30
    if net == "inception" and prec == "f32":
        t1, tm, t2, th, tz = (eng.tensor(w).double() for w in (N.T_ACT1, N.T_MID, N.T_ACT2, N.T_H, N.T_Z))
        for s in range(S):
            cap, pred = capture(cfg, st, x, noise[s], cfg.mode)
            relu = torch.relu
            def cl(name):  # [B,C,L] -> [B*L, C]
                return relu(cap[name]).permute(0, 2, 1).reshape(B * L, -1)
            rows = slice(s * B * L, (s + 1) * B * L)
            for bi, nm in enumerate(["layers.0.conv1.0", "layers.0.conv3.0", "layers.0.conv5.0", "layers.0.convpool.1"]):
                print(f"   s{s} act1[{nm}] relerr {rel_l2(t1[rows, 32 * bi:32 * bi + 27], cl(nm)):.3e}  pad {float(t1[rows, 32 * bi + 27:32 * bi + 32].abs().max()):.1e}")
            print(f"   s{s} mid[b2.0] {rel_l2(tm[rows, :64], cl('layers.1.branch2.0')):.3e} mid[b3.0] {rel_l2(tm[rows, 64:], cl('layers.1.branch3.0')):.3e}")
            for nm, (a, b) in {"layers.1.branch1.0": (0, 16), "layers.1.branch2.2": (16, 32), "layers.1.branch3.2": (32, 48), "layers.1.branch4.1": (48, 80)}.items():
                print(f"   s{s} act2[{nm}] {rel_l2(t2[rows, a:b], cl(nm)):.3e}")
            r2 = slice(s * B, (s + 1) * B)
            print(f"   s{s} h {rel_l2(th[r2], relu(cap['layers.3'])):.3e}  z {rel_l2(tz[r2], cap['last']):.3e}")
    g = eng.grad.cpu()
    for sname, off, num in eng.sites:
        e1 = rel_l2(g[off:off + num], st.mu[sname].grad)
        e2 = rel_l2(g[eng.P + off:eng.P + off + num], st.rho[sname].grad)
        print(f"   grad {sname:28s} mu {e1:.3e} rho {e2:.3e}   |g_mu| {float(st.mu[sname].grad.norm()):.3e} |g_rho| {float(st.rho[sname].grad.norm()):.3e}")
    del eng
    torch.cuda.empty_cache()


def timing(mode, prec, S, B, iters=5):
    from bayesrul_amd.engine import AdamHyper
    ps, qs, lr = HYP[mode]
    eng = _engine("inception", mode, prec, S, B)
    eng.init_params(R.init_mu0("inception", 0, torch.float32), qs)
    x, y = synth_batch(B)
    x, y = x.cuda(), y.cuda()
    hyp = AdamHyper(lr=lr)
    for _ in range(2):
        eng.step(x, y, S, N_DATA, 0.0, ps, hyp, seed=1)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        r = eng.step(x, y, S, N_DATA, 0.0, ps, hyp, seed=1)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    print(f"TIMING {mode} {prec} S={S} B={B}: {dt * 1e3:.3f} ms/step  {S * B / dt:.0f} sample-windows/s  loss {float(r[0]):.6g}")
    del eng
    torch.cuda.empty_cache()


if __name__ == "__main__":
    args = sys.argv[1:] or ["lrt", "flipout", "radial", "normal"]
    for prec in ("f32", "bf16x3"):
        for mode in args:
            if mode in HYP:
                try:
                    run("inception", mode, prec)
                except Exception as e:  # keep going: one log for everything
                    print("FAILED", mode, prec, repr(e))
    if "linear" in args or len(sys.argv) == 1:
        for mode in ("lrt", "flipout"):
            try:
                run("linear", mode, "f32")
            except Exception as e:
                print("FAILED linear", mode, repr(e))
    if "timing" in args or len(sys.argv) == 1:
        for prec in ("bf16x3", "f32"):
            for mode in ("flipout", "lrt", "radial"):
                try:
                    timing(mode, prec, 10, 1000)
                except Exception as e:
                    print("FAILED timing", mode, prec, repr(e))
