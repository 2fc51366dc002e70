"""Synthetic N-CMAPSS-shaped windows: x ~ N(0,1) [n,30,18] (the layout
`NCMAPSSLmdbDataset.__getitem__` yields, data/ncmapss/dataset.py:13-16, after the per-position
standardisation of data/lmdb_utils.py:98-120), y ~ U{0..99} (RUL).  The N-CMAPSS files are not
available here (no network); the window store itself is a 'next' row (§8(f))."""
import torch


class SyntheticWindows:
    def __init__(self, n: int, batch_size: int, win_length: int = 30, n_features: int = 18, seed: int = 1234,
                 shuffle: bool = False, learnable: bool = False):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(n, win_length, n_features, generator=g)
        if learnable:  # RUL that actually depends on the window, so a fit can be seen to learn
            self.y = (50 + 20 * self.x[:, :, 0].mean(1) + 10 * self.x[:, -1, 3]).clamp(0, 99)
        else:
            self.y = torch.randint(0, 100, (n,), generator=g).float()
        self.batch_size, self.shuffle, self.g = batch_size, shuffle, g

    def __len__(self):
        return (self.x.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.x.shape[0]
        idx = torch.randperm(n, generator=self.g) if self.shuffle else torch.arange(n)
        for i in range(0, n, self.batch_size):
            j = idx[i:i + self.batch_size]
            yield self.x[j], self.y[j]
