"""Deterministic `Inception` with the reference's constructor and `state_dict` keys
(bayesrul/models/nets/inception.py:142-217).  Only a weight container + CPU/eager forward for
pre-training and checkpoint interop: the variational forward runs in the HIP kernels."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _branch(*mods):
    return nn.Sequential(*mods)


class InceptionModule(nn.Module):
    def __init__(self, n_features, f1, f3, f5, fp, activation, bias=True):
        super().__init__()
        self.conv1 = _branch(nn.Conv1d(n_features, f1, 1, padding="same", bias=bias), activation())
        self.conv3 = _branch(nn.Conv1d(n_features, f3, 3, padding="same", bias=bias), activation())
        self.conv5 = _branch(nn.Conv1d(n_features, f5, 5, padding="same", bias=bias), activation())
        self.convpool = _branch(nn.MaxPool1d(3, 1, 1), nn.Conv1d(n_features, fp, 3, padding="same", bias=bias),
                                activation())

    def forward(self, x):
        return torch.cat([self.conv1(x), self.conv3(x), self.conv5(x), self.convpool(x)], 1)


class InceptionModuleReducDim(nn.Module):
    def __init__(self, n_features, f1, r3, f3, r5, f5, fp, activation, bias=True):
        super().__init__()
        self.branch1 = _branch(nn.Conv1d(n_features, f1, 1, padding="same", bias=bias), activation())
        self.branch2 = _branch(nn.Conv1d(n_features, r3, 1, padding=0, bias=bias), activation(),
                               nn.Conv1d(r3, f3, 3, padding="same", bias=bias), activation())
        self.branch3 = _branch(nn.Conv1d(n_features, r5, 1, padding=0, bias=bias), activation(),
                               nn.Conv1d(r5, f5, 5, padding="same", bias=bias), activation())
        self.branch4 = _branch(nn.MaxPool1d(3, 1, 1), nn.Conv1d(n_features, fp, 1, padding=0, bias=bias), activation())

    def forward(self, x):
        return torch.cat([self.branch1(x), self.branch2(x), self.branch3(x), self.branch4(x)], 1)


class Inception(nn.Module):
    def __init__(self, win_length, n_features, activation="relu", bias=True, dropout=0, out_size=2):
        super().__init__()
        assert n_features == 18, "Inception is defined for 18 features (nets/inception.py:160-162)"
        if activation != "relu":
            raise ValueError("the MI355X path implements the relu network every reference config uses")
        # dropout > 0 (MC-dropout sibling, conf/model/mcd.yaml): the reference appends nn.Dropout(dropout / 4) to every branch
        # of both blocks and nn.Dropout(dropout) behind the hidden layer (inception.py:48-52,119-123,205-207).  They carry no
        # parameters (same state_dict keys); on this path the masks are drawn inside the HIP kernels (bnn_det_step /
        # bnn_det_forward), so the torch container only records the rate - its eager forward() is the eval-mode network.
        if not 0 <= dropout < 1:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        self.win_length, self.n_features, self.out_size, self.dropout = win_length, n_features, out_size, dropout
        self.layers = nn.Sequential(
            InceptionModule(n_features, 27, 27, 27, 27, nn.ReLU, bias),
            InceptionModuleReducDim(108, 16, 64, 16, 64, 16, 32, nn.ReLU, bias),
            nn.Flatten(), nn.Linear(80 * win_length, 64), nn.ReLU())
        self.last = nn.Linear(64, out_size)
        self.thresh = nn.Threshold(1e-9, 1e-9)

    def forward(self, x):
        return self.thresh(F.softplus(self.last(self.layers(x.transpose(2, 1)))))
