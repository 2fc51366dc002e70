"""Minimal stand-ins for the parts of pytorch-lightning 1.9 the reference's BNN touches
(LightningModule hooks, save_hyperparameters / hparams, self.log, Trainer loops).  Lightning is
not installed here and must not be a hard dependency of the step (SURVEY.md §7 'Hard parts').
If `pytorch_lightning` is importable, models/bayesian.py subclasses the real LightningModule."""
from __future__ import annotations

import inspect
from collections import defaultdict
from types import SimpleNamespace
from typing import Any, Dict, Iterable, List, Optional

import torch


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class LightningModuleLite(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._hparams = AttrDict()
        self._logged: Dict[str, List[torch.Tensor]] = defaultdict(list)
        self.trainer = None
        self._device = torch.device("cpu")

    # -- hparams
    def save_hyperparameters(self, logger=False, ignore=()):
        frame = inspect.currentframe().f_back
        args = inspect.getargvalues(frame)
        for k in args.args:
            if k in ("self",) or k in ignore:
                continue
            self._hparams[k] = args.locals[k]

    @property
    def hparams(self):
        return self._hparams

    @property
    def device(self):
        return self._device

    def to(self, device):
        self._device = torch.device(device)
        return super().to(device)

    # -- logging: tensors are kept on device, reduced once per epoch (no per-step host sync)
    def log(self, name: str, value: Any, on_step: bool = False, on_epoch: bool = True, **kw):
        v = value if torch.is_tensor(value) else torch.tensor(float(value))
        self._logged[name].append(v.detach().float().reshape(()))

    def collect_logs(self) -> Dict[str, float]:
        out = {k: float(torch.stack([t.to("cpu") for t in v]).mean()) for k, v in self._logged.items() if v}
        self._logged.clear()
        return out


class _Progress:
    """optim_step_progress of Lightning's manual-optimisation loop (what bayesian.py:144,156 pokes)."""

    def __init__(self):
        self.ready = self.completed = 0

    def increment_ready(self):
        self.ready += 1

    def increment_completed(self):
        self.completed += 1


class Trainer:
    """fit / validate / test / predict loops over the BNN hooks (single device, like the
    reference's `devices: 1`, conf/trainer/default.yaml:8-12)."""

    def __init__(self, max_epochs: int = 1, device: str = "cuda:0", limit_batches: Optional[int] = None,
                 monitor: str = "elbo/val", patience: Optional[int] = None):
        self.max_epochs, self.device, self.limit = max_epochs, torch.device(device), limit_batches
        self.monitor, self.patience = monitor, patience
        self.history: List[Dict[str, float]] = []
        self.global_step = 0
        progress = _Progress()
        self.fit_loop = SimpleNamespace(epoch_loop=SimpleNamespace(batch_loop=SimpleNamespace(
            manual_loop=SimpleNamespace(optim_step_progress=progress))))

    def _batches(self, loader: Iterable):
        for i, batch in enumerate(loader):
            if self.limit is not None and i >= self.limit:
                break
            yield i, tuple(t.to(self.device, non_blocking=True) for t in batch)

    def fit(self, model, train_loader, val_loader=None):
        model.trainer = self
        model.to(self.device)
        model.on_fit_start()
        best, bad = float("inf"), 0
        for epoch in range(self.max_epochs):
            for i, batch in self._batches(train_loader):
                model.training_step(batch, i)
                self.global_step += 1
            if val_loader is not None:
                for i, batch in self._batches(val_loader):
                    model.validation_step(batch, i)
                if hasattr(model, "validation_epoch_end"):
                    model.validation_epoch_end(None)
            logs = model.collect_logs()
            logs["epoch"] = epoch
            self.history.append(logs)
            if self.patience is not None and self.monitor in logs:
                if logs[self.monitor] < best:
                    best, bad = logs[self.monitor], 0
                else:
                    bad += 1
                    if bad > self.patience:
                        break
        return self.history

    def test(self, model, loader):
        model.trainer = self
        model.to(self.device)
        model.on_test_start()
        for i, batch in self._batches(loader):
            model.test_step(batch, i)
        return model.collect_logs()

    def predict(self, model, loader):
        model.trainer = self
        model.to(self.device)
        model.on_predict_start()
        return [model.predict_step(batch, i) for i, batch in self._batches(loader)]

    # -- checkpoints (Lightning .ckpt shape: state_dict + module hooks)
    def save_checkpoint(self, model, path: str):
        ckpt = {"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "hyper_parameters": {k: v for k, v in model.hparams.items() if not callable(v)},
                "global_step": self.global_step}
        model.on_save_checkpoint(ckpt)
        torch.save(ckpt, path)

    def load_checkpoint(self, model, path: str):
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        model.on_load_checkpoint(ckpt)
        model.load_state_dict(ckpt["state_dict"], strict=False)
        return ckpt
