"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's per-item window read, batched.

Follows `LmdbDataset.__getitem__` (bayesrul/data/lmdb_utils.py:184-194: `np.frombuffer(...).reshape(n_features, -1).T`)
and `NCMAPSSLmdbDataset.__getitem__` (bayesrul/data/ncmapss/dataset.py:13-16: `(sample.copy(), rul)`), collated the way
a DataLoader stacks items.  Pinned by tests/test_window_store.py against the reference's own read expression on
synthetic buffers (no N-CMAPSS LMDB exists here; the `lmdb` package is not installed).
Only tests/ may import this module.
"""
import numpy as np


def gather_windows_ref(x_all: np.ndarray, y_all, idx: np.ndarray, win_length: int, n_features: int,
                       feature_major: bool):
    """x_all: [N, win_length * n_features] fp32 stored windows; returns (x [B, W, F], y [B] or None)."""
    out = np.empty((len(idx), win_length, n_features), dtype=np.float32)
    for i, j in enumerate(idx):
        buf = x_all[j]
        if feature_major:
            out[i] = buf.reshape(n_features, -1).T          # lmdb_utils.py:190-191
        else:
            out[i] = buf.reshape(win_length, n_features)
    y = None if y_all is None else np.asarray(y_all, dtype=np.float32)[idx]
    return out, y
