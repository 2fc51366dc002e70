"""Host orchestrator of the MI355X SVI/ELBO path: owns the flat (mu, rho) parameter buffers,
Adam state and workspace (torch tensors = device memory only) and drives the C ABI.

Replaces, for the hot path, what the reference builds in ``BNN.define_bnn`` /
``BNN.on_fit_start`` out of Pyro + TyXe objects (bayesrul/models/bayesian.py:45-132) and
what ``svi.step`` / ``svi.evaluate_loss`` / ``bnn.predict`` execute per batch (:134-250).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _native as N
from .models.nets.spec import net_layers, site_shapes

NETS = {"inception": N.NET_INCEPTION, "linear": N.NET_LINEAR}
PRECS = {"f32": N.PREC_F32, "bf16x3": N.PREC_BF16X3}


def mode_of(guide: str, fit_context: Optional[str]) -> int:
    """guide / fit_context strings of the reference (bayesian.py:66-85) -> estimator."""
    if guide == "radial":
        return N.MODE_RADIAL  # radial forces the null context (bayesian.py:83)
    if guide != "normal":
        raise RuntimeError("Guide unknown. Choose from 'normal', 'radial'.")
    if fit_context == "lrt":
        return N.MODE_LRT
    if fit_context == "flipout":
        return N.MODE_FLIPOUT
    return N.MODE_NORMAL


@dataclass
class AdamHyper:
    """pyro.optim.ClippedAdam arguments (bayesrul/conf/model/bnn.yaml:6-10)."""
    lr: float = 1e-4
    betas: Tuple[float, float] = (0.95, 0.999)
    eps: float = 1e-8
    clip_norm: float = 15.0
    lrd: float = 1.0
    weight_decay: float = 0.0
    train_loc: bool = True     # guide options of guides/radial.py:74-94 (False: the parameter is left untouched)
    train_scale: bool = True
    torch_eps: bool = False    # torch.optim.Adam's epsilon placement (the frequentist siblings)


@dataclass
class InjectedNoise:
    """Noise handed to the kernels instead of Philox (parity tests).  Layouts are those of
    BnnNoise in include/bayesrul_amd.h; all tensors fp32 on the engine's device."""
    eps_w: Optional[torch.Tensor] = None             # [S, P]
    radial_r: Optional[torch.Tensor] = None          # [S, n_sites]
    lrt_eps: Optional[List[torch.Tensor]] = None     # per layer [S, B, L, Cout] / [S, B, Cout]
    sign_in: Optional[List[torch.Tensor]] = None     # per layer [S, B, cin_img]
    sign_out: Optional[List[torch.Tensor]] = None    # per layer [S, B, Cout]


class SviEngine:
    RESULT_SLOTS = 256   # ring of (loss, kl, loglik) result slots of step(): see its docstring

    def __init__(self, net: str = "inception", guide: str = "normal", fit_context: Optional[str] = "lrt",
                 prec: str = "f32", max_particles: int = 1, max_batch: int = 100, win_length: int = 30,
                 n_features: int = 18, device: str | torch.device = "cuda:0", max_windows: int = 0):
        self.lib = N.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.NativeError("SviEngine needs a HIP device (no CPU fallback)")
        if not torch.cuda.is_available():
            raise N.NativeError("no HIP device visible (torch.cuda.is_available() is False)")
        self.net, self.guide, self.fit_context, self.prec = net, guide, fit_context, prec
        self.mode = mode_of(guide, fit_context)
        self.win_length, self.n_features = win_length, n_features
        self.max_particles, self.max_batch = max_particles, max_batch
        desc = N.PlanDesc(NETS[net], self.mode, PRECS[prec], max_particles, max_batch, win_length, n_features,
                          max_windows)
        self._plan = C.c_void_p()
        N.check(self.lib.bnn_plan_create(C.byref(desc), C.byref(self._plan)))
        P = C.c_int64()
        N.check(self.lib.bnn_plan_num_params(self._plan, C.byref(P)))
        self.P = int(P.value)
        n = C.c_int32()
        N.check(self.lib.bnn_plan_num_sites(self._plan, C.byref(n)))
        self.n_sites = int(n.value)
        N.check(self.lib.bnn_plan_num_layers(self._plan, C.byref(n)))
        self.n_layers = int(n.value)
        self.sites: List[Tuple[str, int, int]] = []
        for i in range(self.n_sites):
            name, off, num = C.c_char_p(), C.c_int64(), C.c_int64()
            N.check(self.lib.bnn_plan_site(self._plan, i, C.byref(name), C.byref(off), C.byref(num)))
            self.sites.append((name.value.decode(), int(off.value), int(num.value)))
        self.layers: List[Tuple[str, int, int, bool]] = []
        for i in range(self.n_layers):
            name, ci, co, cv = C.c_char_p(), C.c_int32(), C.c_int32(), C.c_int32()
            N.check(self.lib.bnn_plan_layer(self._plan, i, C.byref(name), C.byref(ci), C.byref(co), C.byref(cv)))
            self.layers.append((name.value.decode(), int(ci.value), int(co.value), bool(cv.value)))
        wb = C.c_size_t()
        N.check(self.lib.bnn_plan_workspace_bytes(self._plan, C.byref(wb)))
        self.workspace_bytes = int(wb.value)
        dev = self.device
        with torch.cuda.device(dev):
            self.mu = torch.zeros(self.P, dtype=torch.float32, device=dev)
            self.rho = torch.zeros(self.P, dtype=torch.float32, device=dev)
            self.adam_m = torch.zeros(2 * self.P, dtype=torch.float32, device=dev)
            self.adam_v = torch.zeros(2 * self.P, dtype=torch.float32, device=dev)
            self.grad = torch.zeros(2 * self.P + 2, dtype=torch.float32, device=dev)
            self._ws = torch.empty(self.workspace_bytes + 256, dtype=torch.uint8, device=dev)
            base = self._ws.data_ptr()
            self._ws_ptr = (base + 255) // 256 * 256
            self._scal = torch.zeros(4, dtype=torch.float32, device=dev)
            # results of step(): a ring of slots, so that returning (loss, kl, loglik) costs no copy kernel
            self._res_ring = torch.zeros(self.RESULT_SLOTS, 4, dtype=torch.float32, device=dev)
            self._res_k = 0
            bufs = N.Buffers(self.mu.data_ptr(), self.rho.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                             self.grad.data_ptr(), self._ws_ptr, self.workspace_bytes)
            N.check(self.lib.bnn_plan_bind(self._plan, C.byref(bufs)))
        self._shapes = dict(site_shapes(net, win_length, n_features))
        assert [s[0] for s in self.sites] == list(self._shapes), "native / host site tables disagree"
        assert all(math.prod(self._shapes[n]) == num for n, _, num in self.sites)
        self.t = 0            # optimiser steps taken
        self.lr = None        # current (decayed) learning rate
        self._keep = []       # keeps ctypes arrays / tensors alive during a call

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                self.lib.bnn_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    # ---------------------------------------------------------------- parameters
    def site_shape(self, name: str) -> Tuple[int, ...]:
        return self._shapes[name]

    def _view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        for sn, off, num in self.sites:
            if sn == name:
                return buf[off:off + num].view(self.site_shape(name))
        raise KeyError(name)

    def loc(self, name: str) -> torch.Tensor:
        return self._view(self.mu, name)

    def log_scale(self, name: str) -> torch.Tensor:
        return self._view(self.rho, name)

    def init_params(self, mu0: Dict[str, torch.Tensor], q_scale: float) -> None:
        """mu <- net weights, rho <- log(init_scale)  (A3; guides/radial.py:74-95)."""
        with torch.no_grad():
            for sn, off, num in self.sites:
                self.mu[off:off + num].copy_(mu0[sn].reshape(-1).to(self.device, torch.float32))
            self.rho.fill_(math.log(q_scale))
            self.adam_m.zero_()
            self.adam_v.zero_()
        self.t = 0
        self.lr = None

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {"mu": self.mu.detach().cpu(), "rho": self.rho.detach().cpu(), "adam_m": self.adam_m.detach().cpu(),
                "adam_v": self.adam_v.detach().cpu(), "t": torch.tensor(self.t),
                "lr": torch.tensor(-1.0 if self.lr is None else self.lr, dtype=torch.float64)}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        with torch.no_grad():
            self.mu.copy_(sd["mu"].to(self.device))
            self.rho.copy_(sd["rho"].to(self.device))
            if "adam_m" in sd:
                self.adam_m.copy_(sd["adam_m"].to(self.device))
                self.adam_v.copy_(sd["adam_v"].to(self.device))
            self.t = int(sd.get("t", torch.tensor(0)))
            lr = float(sd.get("lr", torch.tensor(-1.0)))
            self.lr = None if lr < 0 else lr

    # ---------------------------------------------------------------- channel maps
    def cin_image_index(self, layer: int) -> torch.Tensor:
        """canonical input-channel index -> image channel index of layer `layer` (the order
        an injected ``sign_in`` row must have; mirrors map_cin in csrc/kernels_misc.h)."""
        name = self.layers[layer][0]
        cin = self._shapes[name + ".weight"][1]
        k = torch.arange(cin)
        if self.net == "inception" and name in ("layers.1.branch1.0", "layers.1.branch2.0", "layers.1.branch3.0",
                                                "layers.1.branch4.1"):
            return (k // 27) * 32 + k % 27          # block-1 branch outputs are stored 27 -> 32 padded
        if self.net == "inception" and name == "layers.3":
            Lw = self.win_length
            return (k % Lw) * 80 + k // Lw          # nn.Flatten of [C=80, L]: c*L + l -> l*80 + c
        return k

    # ---------------------------------------------------------------- call plumbing
    def _stream(self) -> int:
        return int(torch.cuda.current_stream(self.device).cuda_stream)

    def _noise(self, noise: Optional[InjectedNoise], seed: int, step: int) -> N.Noise:
        nz = N.Noise()
        nz.seed, nz.step = seed, step
        self._keep = []
        if noise is None:
            return nz

        def chk(t):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "noise must be fp32 contiguous on device"
            self._keep.append(t)
            return t.data_ptr()

        def arr(lst):
            a = (C.c_void_p * self.n_layers)()
            for i in range(self.n_layers):
                a[i] = chk(lst[i]) if lst[i] is not None else None
            self._keep.append(a)
            return C.cast(a, C.POINTER(C.c_void_p))

        if noise.eps_w is not None:
            nz.eps_w = chk(noise.eps_w)
        if noise.radial_r is not None:
            nz.radial_r = chk(noise.radial_r)
        if noise.lrt_eps is not None:
            nz.lrt_eps = arr(noise.lrt_eps)
        if noise.sign_in is not None:
            nz.sign_in = arr(noise.sign_in)
        if noise.sign_out is not None:
            nz.sign_out = arr(noise.sign_out)
        return nz

    def _elbo_args(self, x, y, S, dataset_size, prior_loc, prior_scale, mode, with_obs, scaled, goff, gbatch):
        a = N.ElboArgs()
        if x is not None:
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            assert tuple(x.shape[1:]) == (self.win_length, self.n_features), f"x must be [B,{self.win_length},{self.n_features}]"
            a.x = x.data_ptr()
            a.batch = x.shape[0]
        if y is not None:
            assert y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and y.numel() == x.shape[0]
            a.y = y.data_ptr()
        if x is None:
            a.batch = 1
        a.particles = S
        a.global_batch_offset = goff
        a.global_batch = gbatch if gbatch else a.batch
        a.dataset_size = float(dataset_size)
        a.prior_loc, a.prior_scale = float(prior_loc), float(prior_scale)
        a.mode_override = -1 if mode is None else mode
        a.with_obs, a.scaled = int(with_obs), int(scaled)
        return a

    def _adam_args(self, h: AdamHyper, grad_scale: float) -> N.AdamArgs:
        if self.lr is None:
            self.lr = h.lr
        self.t += 1
        self.lr *= h.lrd  # pyro ClippedAdam decays lr at the top of step()
        return N.AdamArgs(self.lr, h.betas[0], h.betas[1], h.eps, h.clip_norm, h.weight_decay, self.t, grad_scale,
                          int(not h.train_loc), int(not h.train_scale), int(h.torch_eps), 0)

    # ---------------------------------------------------------------- the path
    def step(self, x: torch.Tensor, y: torch.Tensor, particles: int, dataset_size: float, prior_loc: float,
             prior_scale: float, adam: Optional[AdamHyper], noise: Optional[InjectedNoise] = None, seed: int = 0,
             step: Optional[int] = None, want_preds: bool = False, global_batch: int = 0,
             global_batch_offset: int = 0, keep: bool = True):
        """svi.step (A4).  Returns (loss, kl, loglik) as a 3-element device tensor (no host
        sync) and optionally preds [S,B,2].  With ``adam=None`` the gradient is left in
        ``self.grad`` for the DP all-reduce (see parallel.py) and no update is applied.
        ``keep=True`` (default) returns a private copy of the three scalars; ``keep=False`` returns a VIEW of one of
        ``RESULT_SLOTS`` ring slots (no copy kernel) that stays valid for the next RESULT_SLOTS - 1 steps only - for
        callers that read the result at once (bench.py, parallel.dp_step)."""
        with torch.cuda.device(self.device):
            a = self._elbo_args(x, y, particles, dataset_size, prior_loc, prior_scale, None, 1, 1,
                                global_batch_offset, global_batch)
            nz = self._noise(noise, seed, self.t if step is None else step)
            preds = torch.empty(particles, x.shape[0], 2, dtype=torch.float32, device=self.device) if want_preds else None
            slot = self._res_ring[self._res_k]
            self._res_k = (self._res_k + 1) % self.RESULT_SLOTS
            sp = slot.data_ptr()
            out = N.ElboOut(sp, sp + 4, sp + 8, N.ptr(preds))
            ad = self._adam_args(adam, 1.0) if adam is not None else None
            N.check(self.lib.bnn_elbo_step(self._plan, C.byref(a), C.byref(nz), C.byref(ad) if ad else None,
                                           C.byref(out), C.c_void_p(self._stream())))
            res = slot[:3].clone() if keep else slot[:3]
        return (res, preds) if want_preds else res

    def _dropout(self, p: float, seed: int, step: int, keep=None):
        """BnnDropout of one call: rate p, Philox (seed, step) or injected keep masks (act1 [B*W,128], act2 [B*W,80], h [B,64])."""
        d = N.Dropout(float(p), int(seed), int(step), 0, 0, 0)
        if keep is not None:
            k1, k2, kh = (t.to(self.device, torch.float32).contiguous() for t in keep)
            d.keep_act1, d.keep_act2, d.keep_h = k1.data_ptr(), k2.data_ptr(), kh.data_ptr()
            d._keep = (k1, k2, kh)   # keep the buffers alive for the call
        return d

    def det_forward(self, x: torch.Tensor, dropout: Optional["N.Dropout"] = None) -> torch.Tensor:
        """net(x) with weights = mu -> [B, 2] (loc, scale); `dropout` (from `_dropout`) keeps nn.Dropout active: one pass of
        HNN.mc_sampling (frequentist.py:60-81)."""
        with torch.cuda.device(self.device):
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            preds = torch.empty(x.shape[0], 2, dtype=torch.float32, device=self.device)
            N.check(self.lib.bnn_det_forward(self._plan, C.c_void_p(x.data_ptr()), C.c_int32(x.shape[0]),
                                             C.byref(dropout) if dropout is not None else None,
                                             C.c_void_p(preds.data_ptr()), C.c_void_p(self._stream())))
        return preds

    def det_step(self, x: torch.Tensor, y: torch.Tensor, objective: str, adam: Optional[AdamHyper],
                 want_preds: bool = True, dropout: Optional["N.Dropout"] = None):
        """One deterministic training step of the net with weights = mu (the frequentist siblings, SURVEY.md 8(f) rank
        4): `objective` = "gaussian_nll" (HNN.step, frequentist.py:39-48) or "mse" (NN.step, :173-178).  Returns the
        mean loss as a 1-element device tensor and the net outputs [B, 2]; d loss / d mu is left in grad[:P]."""
        obj = {"gaussian_nll": 1, "mse": 2}[objective]
        with torch.cuda.device(self.device):
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and y.is_cuda and y.dtype == torch.float32
            B = x.shape[0]
            d = N.DetArgs(x.data_ptr(), y.data_ptr(), B, obj, C.pointer(dropout) if dropout is not None else None)
            preds = torch.empty(1, B, 2, dtype=torch.float32, device=self.device) if want_preds else None
            out = N.ElboOut(self._scal.data_ptr(), self._scal.data_ptr() + 4, self._scal.data_ptr() + 8, N.ptr(preds))
            ad = self._adam_args(adam, 1.0) if adam is not None else None
            N.check(self.lib.bnn_det_step(self._plan, C.byref(d), C.byref(ad) if ad else None, C.byref(out),
                                          C.c_void_p(self._stream())))
            loss = self._scal[:1].clone()
        return loss, (preds[0] if want_preds else None)

    def apply_adam(self, adam: AdamHyper, grad_scale: float = 1.0) -> None:
        with torch.cuda.device(self.device):
            ad = self._adam_args(adam, grad_scale)
            N.check(self.lib.bnn_clipped_adam(self._plan, C.byref(ad), C.c_void_p(self._stream())))

    def evaluate(self, x: Optional[torch.Tensor], y: Optional[torch.Tensor], particles: int, dataset_size: float,
                 prior_loc: float, prior_scale: float, mode: Optional[int] = None, with_obs: bool = True,
                 scaled: bool = True, noise: Optional[InjectedNoise] = None, seed: int = 0, step: int = 0,
                 want_preds: bool = False):
        """svi.evaluate_loss (A4/A15).  mode=None keeps the training estimator; validation in
        the reference runs outside fit_ctxt, i.e. plain sampling: pass MODE_NORMAL / MODE_RADIAL."""
        with torch.cuda.device(self.device):
            a = self._elbo_args(x, y, particles, dataset_size, prior_loc, prior_scale, mode, with_obs, scaled, 0, 0)
            nz = self._noise(noise, seed, step)
            preds = (torch.empty(particles, x.shape[0], 2, dtype=torch.float32, device=self.device)
                     if (want_preds and x is not None) else None)
            out = N.ElboOut(self._scal.data_ptr(), self._scal.data_ptr() + 4, self._scal.data_ptr() + 8, N.ptr(preds))
            N.check(self.lib.bnn_elbo_evaluate(self._plan, C.byref(a), C.byref(nz), C.byref(out),
                                               C.c_void_p(self._stream())))
            res = self._scal[:3].clone()
        return (res, preds) if want_preds else res

    def plain_mode(self) -> int:
        return N.MODE_RADIAL if self.mode == N.MODE_RADIAL else N.MODE_NORMAL

    def predict(self, x: torch.Tensor, particles: int, noise: Optional[InjectedNoise] = None, seed: int = 0,
                step: int = 0, want_samples: bool = True):
        """bnn.predict(x, num_predictions=S, aggregate=False) + predict_step aggregation (A16).
        Returns (out4 [4,B] = preds, stds, ep_vars, al_vars; samples [S,B,2] or None)."""
        with torch.cuda.device(self.device):
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            B = x.shape[0]
            nz = self._noise(noise, seed, step)
            samples = torch.empty(particles, B, 2, dtype=torch.float32, device=self.device) if (
                want_samples or particles > self.max_particles) else None
            out4 = torch.empty(4, B, dtype=torch.float32, device=self.device)
            N.check(self.lib.bnn_predict(self._plan, C.c_void_p(x.data_ptr()), B, particles, C.byref(nz),
                                         C.c_void_p(N.ptr(samples)), C.c_void_p(out4.data_ptr()),
                                         C.c_void_p(self._stream())))
        return out4, samples

    def export_noise(self, batch: int, particles: int, seed: int, step: int, mode: Optional[int] = None,
                     global_batch: int = 0, global_batch_offset: int = 0) -> InjectedNoise:
        """The Philox noise the kernels draw for (seed, step), in injectable layouts."""
        mode = self.mode if mode is None else mode
        dev, S, B, Lw = self.device, particles, batch, self.win_length
        out = InjectedNoise()
        with torch.cuda.device(dev):
            a = self._elbo_args(None, None, S, 1.0, 0.0, 1.0, mode, 0, 0, global_batch_offset, global_batch or batch)
            a.batch = batch
            a.global_batch = global_batch or batch
            out.eps_w = torch.empty(S, self.P, dtype=torch.float32, device=dev)
            out.radial_r = torch.empty(S, self.n_sites, dtype=torch.float32, device=dev)
            mk = lambda lst: (C.c_void_p * self.n_layers)(*[t.data_ptr() for t in lst])
            out.lrt_eps = [torch.empty((S, B, Lw, co) if conv else (S, B, co), dtype=torch.float32, device=dev)
                           for (_, ci, co, conv) in self.layers]
            out.sign_in = [torch.empty(S, B, ci, dtype=torch.float32, device=dev) for (_, ci, co, conv) in self.layers]
            out.sign_out = [torch.empty(S, B, co, dtype=torch.float32, device=dev) for (_, ci, co, conv) in self.layers]
            a1, a2, a3 = mk(out.lrt_eps), mk(out.sign_in), mk(out.sign_out)
            N.check(self.lib.bnn_export_noise(self._plan, C.byref(a), C.c_uint64(seed), C.c_uint64(step),
                                              C.c_void_p(out.eps_w.data_ptr()), C.c_void_p(out.radial_r.data_ptr()),
                                              a1, a2, a3, C.c_void_p(self._stream())))
        return out

    KINDS = ["fwd", "dx", "dw", "sample", "head", "finalize", "adam", "pool_bwd", "noise"]

    def profile(self, on: bool, only=None) -> None:
        """Per-kernel HIP-event timing on the launch stream; `only` = iterable of (kind, group) to
        restrict the recorder to (two events per recorded launch sit between the kernels)."""
        tags = [self.KINDS.index(k) * 16 + g for (k, g) in (only or [])]
        arr = (C.c_int32 * max(1, len(tags)))(*tags)
        N.check(self.lib.bnn_profile_select(self._plan, arr, len(tags)))
        N.check(self.lib.bnn_profile_enable(self._plan, int(on)))

    def profile_symbol(self, kind: str, group: int) -> str:
        """Kernel symbol last recorded under (kind, group), as rocprofv3 prints it (no argument list)."""
        buf = C.create_string_buffer(200)
        N.check(self.lib.bnn_profile_name(self._plan, self.KINDS.index(kind) * 16 + group, buf, 200))
        return buf.value.decode()

    def profile_read(self) -> Dict[Tuple[str, int], Tuple[float, int]]:
        """{(kind, group): (total ms, launches)} since the last read (HIP events on the stream)."""
        cap = 256
        tags, ms, cnt, n = (C.c_int32 * cap)(), (C.c_double * cap)(), (C.c_int64 * cap)(), C.c_int32()
        N.check(self.lib.bnn_profile_read(self._plan, tags, ms, cnt, cap, C.byref(n)))
        kinds = self.KINDS
        return {(kinds[tags[i] // 16], tags[i] % 16): (ms[i], cnt[i]) for i in range(n.value)}

    def tensor(self, which: int) -> torch.Tensor:
        """Copy of an intermediate activation of the last forward (tests)."""
        p, rows, ct = C.c_void_p(), C.c_int64(), C.c_int32()
        N.check(self.lib.bnn_plan_tensor(self._plan, which, C.byref(p), C.byref(rows), C.byref(ct)))
        n = rows.value * ct.value
        off = (p.value - self._ws.data_ptr())
        torch.cuda.synchronize(self.device)
        return self._ws[off:off + 4 * n].view(torch.float32).view(rows.value, ct.value).clone()
