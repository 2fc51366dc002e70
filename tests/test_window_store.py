"""Window store: HBM-resident windows, one gather launch per batch."""
import numpy as np
import pytest
import torch

from oracle.window_store import gather_windows_ref


def test_oracle_follows_the_reference_read_expression():
    """The restatement against the literal expression of LmdbDataset.__getitem__ on a raw LMDB-style buffer."""
    rng = np.random.default_rng(0)
    W, F, N = 30, 18, 7
    raw = [rng.standard_normal(W * F).astype(np.float32).tobytes() for _ in range(N)]     # what txn.get returns
    ruls = [str(float(i * 3)) for i in range(N)]                                           # 'rul_{i}' decimal strings
    x_all = np.stack([np.frombuffer(b, dtype=np.float32) for b in raw])
    idx = np.array([5, 0, 0, 6])
    x, y = gather_windows_ref(x_all, [np.float32(r) for r in ruls], idx, W, F, feature_major=True)
    for i, j in enumerate(idx):
        sample = np.frombuffer(raw[j], dtype=np.float32)
        sample = sample.reshape(F, -1).T                                                   # lmdb_utils.py:189-190
        assert np.array_equal(x[i], sample) and y[i] == np.float32(ruls[j])
    assert x.shape == (4, W, F)


@pytest.mark.gpu
@pytest.mark.parametrize("feature_major", [False, True])
def test_device_store_matches_oracle_and_covers_an_epoch(feature_major):
    from bayesrul_amd.data.window_store import DeviceWindowStore
    g = torch.Generator().manual_seed(3)
    N, W, F, B = 1003, 30, 18, 100          # ragged last batch
    x_all = torch.randn(N, W * F, generator=g)
    y_all = torch.randint(0, 100, (N,), generator=g).float()
    store = DeviceWindowStore(x_all, y_all, batch_size=B, win_length=W, n_features=F, feature_major=feature_major,
                              shuffle=True, seed=11)
    idx = torch.tensor([0, N - 1, 17, 17, 512])
    xo, yo = store.gather(idx.cuda())
    xr, yr = gather_windows_ref(x_all.numpy(), y_all.numpy(), idx.numpy(), W, F, feature_major)
    assert np.array_equal(xo.cpu().numpy(), xr) and np.array_equal(yo.cpu().numpy(), yr)     # bit-exact copy
    # an epoch visits every window exactly once (labels carry the index)
    store.y = torch.arange(N, dtype=torch.float32, device="cuda")
    seen, nb = [], 0
    for xb, yb in store:
        nb += 1
        assert xb.shape[1:] == (W, F) and xb.shape[0] == yb.shape[0] <= B
        seen.append(yb.cpu())
    assert nb == len(store) == 11
    seen = torch.cat(seen).long()
    assert torch.equal(seen.sort().values, torch.arange(N))
    assert not torch.equal(seen, torch.arange(N))            # shuffled
    xe, ye = store.gather(torch.empty(0, dtype=torch.int64, device="cuda"))
    assert xe.shape == (0, W, F)
