"""`HNN` and `NN` — drop-ins for `bayesrul.models.frequentist.HNN / NN` (bayesrul/models/frequentist.py:9-188), the
frequentist siblings of the BNN (BASELINE config[0] `experiment=ncmapss_hnn`).

Same constructor keywords, hook names, logged keys and `predict_step` dictionary.  The training step runs on the same
HIP kernels as the variational path, with weights = mu and no sampling (`bnn_det_step`): deterministic forward ->
Gaussian NLL (`HNN`) or MSE (`NN`) -> backward -> torch.optim.Adam semantics (L2 weight decay added to the gradient,
epsilon outside the bias-corrected square root).  Differences, all deliberate:
  * `optimizer` is the Adam argument dict of conf/model/nn.yaml:6-10 (`{lr, weight_decay[, betas, eps]}`) or a
    `functools.partial(torch.optim.Adam, ...)` whose keywords are read — Lightning's automatic optimisation is replaced
    by the fused device optimiser, `configure_optimizers` returns None;
  * MC-dropout (`net.dropout > 0`, conf/experiment/ncmapss_mcd.yaml): the keep masks of the nine nn.Dropout modules are
    drawn inside the kernels from a Philox stream (seed, step counter) instead of torch's generator; the training step
    runs with dropout active, validation / test / predict run `mc_samples` stochastic passes (`mc_sampling`) as the
    reference does.  Exact-fp32 plan only (`prec="f32"`);
  * `validation_epoch_end` (Lightning 1.9 API) is kept and also fed by the lite trainer.
There is no CPU fallback: the step needs the HIP library and a gfx950 device.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from ..engine import AdamHyper, SviEngine
from ..results.metrics import rms_calibration_error, sharpness
from ..utils.miscellaneous import weights_init

try:  # pragma: no cover - Lightning is optional
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover
    from ..lightning_lite import LightningModuleLite as _Base


def adam_hyper_of(optimizer) -> AdamHyper:
    """torch.optim.Adam arguments -> the fused optimiser's (no clamp, torch's epsilon placement, rho frozen)."""
    if isinstance(optimizer, AdamHyper):
        return optimizer
    kw = dict(getattr(optimizer, "keywords", None) or optimizer)
    return AdamHyper(lr=kw.get("lr", 1e-3), betas=tuple(kw.get("betas", (0.9, 0.999))), eps=kw.get("eps", 1e-8),
                     clip_norm=float("inf"), weight_decay=kw.get("weight_decay", 0.0), train_scale=False, torch_eps=True)


class _DetModule(_Base):
    """What HNN and NN share: the engine that holds the net's parameters on the device."""

    def __init__(self):
        super().__init__()
        self.engine: Optional[SviEngine] = None

    def _net_kind(self) -> str:
        return "inception" if "inception" in type(self.net).__name__.lower() else "linear"

    def _ensure_engine(self, prec: str, max_batch: int) -> SviEngine:
        if self.engine is None:
            if getattr(self.net, "dropout", 0) > 0 and (prec != "f32" or self._net_kind() != "inception"):
                raise RuntimeError("MC-dropout runs on the exact-fp32 Inception kernels: use prec='f32'")
            self.engine = SviEngine(net=self._net_kind(), guide="normal", fit_context=None, prec=prec, max_particles=1,
                                    max_batch=max_batch, win_length=self.net.win_length,
                                    n_features=self.net.n_features, device=self.device)
            self.engine.init_params({k: v.detach() for k, v in self.net.state_dict().items()}, 1.0)
            self._zero_eps = torch.zeros(1, self.engine.P, dtype=torch.float32, device=self.engine.device)
        return self.engine

    def sync_net(self) -> torch.nn.Module:
        """device parameters -> the torch module (checkpoints, `load_pretrained_net`)."""
        if self.engine is not None:
            with torch.no_grad():
                sd = self.net.state_dict()
                for name, _, _ in self.engine.sites:
                    sd[name].copy_(self.engine.loc(name).to(sd[name].device))
        return self.net

    # ---- MC-dropout: Philox stream of the keep masks = (seed, a counter that advances with every stochastic pass)
    _drop_seed, _drop_count = 0x5EED, 0

    def _next_dropout(self, keep=None):
        """BnnDropout of the next stochastic pass, or None when the net has no dropout."""
        p = float(getattr(self.net, "dropout", 0) or 0)
        if p <= 0:
            return None
        self._drop_count += 1
        return self._ensure_engine(self._prec, self._max_batch)._dropout(p, self._drop_seed, self._drop_count, keep)

    def forward(self, x, dropout=None):
        """net(x) with the current weights, on the device kernels: [B, 2].  `dropout`: a BnnDropout (stochastic pass)."""
        from ..engine import InjectedNoise
        eng = self._ensure_engine(self._prec, self._max_batch)
        x = x.contiguous().float()
        if dropout is not None:
            return torch.cat([eng.det_forward(x[b0:b0 + eng.max_batch].contiguous(), dropout)
                              for b0 in range(0, x.shape[0], eng.max_batch)])
        # the reference's test / predict loaders use test_batch_size = 10000 (data/ncmapss/dataset.py:40,126,135): evaluate in
        # chunks of the engine's batch capacity
        outs = []
        for b0 in range(0, x.shape[0], eng.max_batch):
            _, samples = eng.predict(x[b0:b0 + eng.max_batch].contiguous(), 1, noise=InjectedNoise(eps_w=self._zero_eps))
            outs.append(samples[0])
        return outs[0] if len(outs) == 1 else torch.cat(outs)

    def configure_optimizers(self):
        return None

    def on_save_checkpoint(self, checkpoint: Dict) -> None:
        self.sync_net()
        checkpoint["state_dict"] = {k: v.detach().cpu() for k, v in self.state_dict().items()}


class HNN(_DetModule):
    """bayesrul/models/frequentist.py:9-154"""

    def __init__(self, net: torch.nn.Module, optimizer, mc_samples: int = 0, p_dropout: float = 0, prec: str = "f32",
                 max_batch: int = 1000):
        super().__init__()
        self.save_hyperparameters(logger=False, ignore=["net"])
        self.net = net
        self.net.apply(weights_init)   # frequentist.py:29
        self._prec, self._max_batch = prec, max_batch
        self.adam = adam_hyper_of(optimizer)
        self._val = []

    # ---- frequentist.py:39-48
    def step(self, batch, phase, stochastic: bool = False):
        """frequentist.py:39-48.  `stochastic`: nn.Dropout active (the training step always is: Lightning runs it in train
        mode; `enable_dropout` + `mc_sampling` turn it on for the other phases, frequentist.py:83-92)."""
        x, y = batch[0].contiguous().float(), batch[1].contiguous().float().reshape(-1)
        if phase == "train":
            loss, out = self._ensure_engine(self._prec, self._max_batch).det_step(x, y, "gaussian_nll", self.adam,
                                                                                 dropout=self._next_dropout())
            loss = loss[0]
        else:
            out = self.forward(x, self._next_dropout() if stochastic else None)
            if phase == "predict":
                return out[:, 0], out[:, 1]
            loss = F.gaussian_nll_loss(out[:, 0], y, torch.square(out[:, 1]))
        self.log(f"nll/{phase}", loss, on_step=False, on_epoch=True)
        return loss, out[:, 0], out[:, 1]

    # ---- frequentist.py:50-58
    def training_step(self, batch, batch_idx):
        loss, loc, scale = self.step(batch, "train")
        y = batch[1].float().reshape(-1)
        self.log("mse/train", F.mse_loss(loc, y), on_step=False, on_epoch=True)
        self.log("rmsce/train", rms_calibration_error(loc, scale, y), on_step=False, on_epoch=True)
        self.log("sharp/train", sharpness(scale), on_step=False, on_epoch=True)
        return loss

    # ---- frequentist.py:60-81
    def mc_sampling(self, batch, mc_samples: int, phase: str, agg: bool = True):
        losses, locs, scales = [], [], []
        for _ in range(mc_samples):
            if phase == "predict":
                loc, scale = self.step(batch, phase, stochastic=True)
            else:
                loss, loc, scale = self.step(batch, phase, stochastic=True)
                losses.append(loss)
            locs.append(loc)
            scales.append(scale)
        locs, scales = torch.stack(locs), torch.stack(scales)
        if phase == "predict":
            return locs, scales
        loss = torch.stack(losses).mean(0)
        if agg:
            return loss, locs.mean(0), scales.pow(2).mean(0).add(locs.var(0)).sqrt()
        return loss, locs, scales

    # ---- frequentist.py:83-113
    def validation_step(self, batch, batch_idx):
        if self.net.dropout > 0:
            loss, loc, scale = self.mc_sampling(batch, self.hparams.mc_samples, phase="val")
        else:
            loss, loc, scale = self.step(batch, "val")
        out = {"loss": loss, "label": batch[1].float().reshape(-1), "pred": loc, "std": scale}
        self._val.append(out)
        return out

    def validation_epoch_end(self, outputs=None) -> None:
        outputs = self._val if outputs is None else outputs
        if not outputs:
            return
        preds = torch.cat([o["pred"].detach() for o in outputs])
        labels = torch.cat([o["label"].detach() for o in outputs])
        stds = torch.cat([o["std"].detach() for o in outputs])
        self.log("mse/val", F.mse_loss(preds, labels))
        self.log("rmsce/val", rms_calibration_error(preds, stds, labels))
        self.log("sharp/val", sharpness(stds))
        self._val = []

    # ---- frequentist.py:115-134
    def test_step(self, batch, batch_idx):
        y = batch[1].float().reshape(-1)
        if self.net.dropout > 0:
            loss, locs, scales = self.mc_sampling(batch, self.hparams.mc_samples, phase="test", agg=False)
            ep_var, al_var = locs.var(0), (scales**2).mean(0)
            scale, loc = al_var.add(ep_var).sqrt(), locs.mean(0)
        else:
            loss, loc, scale = self.step(batch, "test")
        self.log("nll/test", loss)
        self.log("mse/test", F.mse_loss(loc, y))
        self.log("rmsce/test", rms_calibration_error(loc, scale, y))
        self.log("sharp/test", sharpness(scale))

    # ---- frequentist.py:136-151
    def predict_step(self, batch, batch_idx, dataloader_idx=0):
        pred = {"labels": batch[1].cpu().numpy()}
        if self.net.dropout > 0:
            locs, scales = self.mc_sampling(batch, self.hparams.mc_samples, phase="predict", agg=False)
            ep_var, al_var = locs.var(0), (scales**2).mean(0)
            scale, loc = al_var.add(ep_var).sqrt(), locs.mean(0)
            pred["ep_vars"], pred["al_vars"] = ep_var.cpu().numpy(), al_var.cpu().numpy()
        else:
            loc, scale = self.step(batch, "predict")
        pred["preds"], pred["stds"] = loc.cpu().numpy(), scale.cpu().numpy()
        return pred

    def on_fit_start(self) -> None:
        self._ensure_engine(self._prec, self._max_batch)


class NN(_DetModule):
    """bayesrul/models/frequentist.py:157-188: MSE pre-training of the net a BNN later starts from
    (`pretrain_epochs > 0`, tasks/train.py:75-77)."""

    def __init__(self, net: torch.nn.Module, optimizer, prec: str = "f32", max_batch: int = 1000):
        super().__init__()
        self.save_hyperparameters(logger=False, ignore=["net"])
        self.net = net
        self.net.apply(weights_init)   # frequentist.py:169
        self._prec, self._max_batch = prec, max_batch
        self.adam = adam_hyper_of(optimizer)

    # ---- frequentist.py:173-178
    def step(self, batch, train: bool = False):
        x, y = batch[0].contiguous().float(), batch[1].contiguous().float().reshape(-1)
        if train:
            loss, _ = self._ensure_engine(self._prec, self._max_batch).det_step(x, y, "mse", self.adam, want_preds=False)
            return loss[0]
        return F.mse_loss(self.forward(x)[:, 0], y)

    def training_step(self, batch, batch_idx):
        return self.step(batch, train=True)

    def validation_step(self, batch, batch_idx):
        self.log("mse/val", self.step(batch), on_step=False, on_epoch=True)

    def on_fit_start(self) -> None:
        self._ensure_engine(self._prec, self._max_batch)


def load_pretrained_net(ckpt_path, net: torch.nn.Module) -> torch.nn.Module:
    """tasks/train.py:106-131: `NN.load_from_checkpoint(ckpt_path, net=model.net).net` — the pre-trained weights of an
    `NN` checkpoint (keys `net.<param>`) loaded into `net`.  Returns `net` untouched when the file is missing, as the
    reference falls back to a freshly instantiated net."""
    import os
    if ckpt_path is None or not os.path.exists(ckpt_path):
        return net
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    sd = {k[len("net."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("net.")}
    net.load_state_dict(sd)
    return net
