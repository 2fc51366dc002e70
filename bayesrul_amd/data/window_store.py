"""HBM-resident window store.

The reference feeds the step through a 12-worker DataLoader over an LMDB
(`bayesrul/data/ncmapss/dataset.py:10-139`, `bayesrul/data/lmdb_utils.py:164-204`): one key per window, value =
fp32 `[n_features][win_length]` transposed on read, label `rul_{i}` stored as a decimal string.  At the
step rates of this package (millions of MC-samples x windows per second) that loader is the bottleneck, and
the whole N-CMAPSS train split is only ~238 k x 540 x 4 B = 514 MB: it is kept in HBM and a batch is one
gather launch (`bnn_gather_windows`); the shuffle is a device permutation.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterator, Optional, Tuple

import torch

from .. import _native as N


class DeviceWindowStore:
    """x: [N, win_length, n_features] fp32 (row-major windows) or, with `feature_major=True`, the raw LMDB values
    [N, n_features * win_length]; y: [N] fp32.  Iterating yields (x [B, W, F], y [B]) device tensors, same
    items as `NCMAPSSLmdbDataset.__getitem__` collated by a DataLoader (drop_last=False)."""

    def __init__(self, x: torch.Tensor, y: Optional[torch.Tensor], batch_size: int, win_length: int = 30,
                 n_features: int = 18, feature_major: bool = False, shuffle: bool = True, seed: int = 0,
                 device: str | torch.device = "cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise N.NativeError("DeviceWindowStore needs a HIP device (no CPU fallback)")
        self.lib = N.load()
        n = x.shape[0]
        if x.numel() != n * win_length * n_features:
            raise ValueError(f"x holds {x.numel()} values, expected {n} x {win_length} x {n_features}")
        self.x = x.to(self.device, torch.float32).contiguous().view(n, -1)
        self.y = None if y is None else y.to(self.device, torch.float32).contiguous()
        self.n, self.W, self.F = n, win_length, n_features
        self.feature_major, self.batch_size, self.shuffle = bool(feature_major), int(batch_size), shuffle
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    @classmethod
    def from_lmdb(cls, path, pattern: str = "{}", **kw) -> "DeviceWindowStore":
        """Loads the reference's LMDB once (keys `pattern.format(i)`, `rul_{i}`, `nb_lines`, `n_features`, `bits`;
        lmdb_utils.py:164-204).  Needs the `lmdb` package, which this image does not ship."""
        try:
            import lmdb  # noqa: F401
        except ImportError as e:  # pragma: no cover - not installable here
            raise ImportError("DeviceWindowStore.from_lmdb needs the 'lmdb' package") from e
        import numpy as np
        env = lmdb.open(str(path), readonly=True)
        with env.begin(write=False) as txn:
            n = int(txn.get(b"nb_lines").decode())
            nf = int(txn.get(b"n_features").decode())
            dt = np.float32 if int(txn.get(b"bits").decode()) == 32 else np.float64
            rows = [np.frombuffer(txn.get(pattern.format(i).encode()), dtype=dt).astype(np.float32) for i in range(n)]
            ruls = [float(txn.get(f"rul_{i}".encode()).decode()) for i in range(n)]
        x = torch.from_numpy(np.stack(rows))
        return cls(x, torch.tensor(ruls), win_length=x.shape[1] // nf, n_features=nf, feature_major=True, **kw)

    def __len__(self) -> int:
        return (self.n + self.batch_size - 1) // self.batch_size

    def gather(self, idx: torch.Tensor) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """One launch: the windows (and labels) at `idx` (int64 device tensor)."""
        idx = idx.to(self.device, torch.int64).contiguous()
        b = idx.numel()
        xo = torch.empty(b, self.W, self.F, dtype=torch.float32, device=self.device)
        yo = None if self.y is None else torch.empty(b, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device).cuda_stream
            N.check(self.lib.bnn_gather_windows(
                C.c_void_p(self.x.data_ptr()), C.c_void_p(N.ptr(self.y)), C.c_void_p(idx.data_ptr()), C.c_int64(b),
                C.c_int64(self.n), C.c_int32(self.W), C.c_int32(self.F), C.c_int32(int(self.feature_major)), C.c_void_p(xo.data_ptr()),
                C.c_void_p(N.ptr(yo)), C.c_void_p(st)))
        return xo, yo

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
        perm = (torch.randperm(self.n, device=self.device, generator=self.gen) if self.shuffle
                else torch.arange(self.n, device=self.device))
        for i in range(0, self.n, self.batch_size):
            yield self.gather(perm[i:i + self.batch_size])
