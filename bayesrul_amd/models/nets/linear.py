"""`Linear` MLP 540->256->128->128->32->out with the reference's `state_dict` keys
(bayesrul/models/nets/linear.py:10-72).  Unlike the reference it carries `win_length` /
`n_features` (which `BNN.on_fit_start` reads, bayesian.py:117-118) and defaults
to `out_size=2` (the BNN needs loc and scale)."""
import torch.nn as nn
import torch.nn.functional as F


class Linear(nn.Module):
    def __init__(self, win_length, n_features, activation="relu", dropout=0, bias=True, out_size=2):
        super().__init__()
        if activation != "relu" or dropout:
            raise ValueError("the MI355X path implements the relu / no-dropout network")
        self.win_length, self.n_features, self.out_size, self.dropout = win_length, n_features, out_size, dropout
        self.layers = nn.Sequential(
            nn.Flatten(), nn.Linear(win_length * n_features, 256, bias=bias), nn.ReLU(),
            nn.Linear(256, 128, bias=bias), nn.ReLU(), nn.Linear(128, 128, bias=bias), nn.ReLU(),
            nn.Linear(128, 32, bias=bias), nn.ReLU())
        self.last = nn.Linear(32, out_size)
        self.thresh = nn.Threshold(1e-9, 1e-9)

    def forward(self, x):
        return self.thresh(F.softplus(self.last(self.layers(x.unsqueeze(1)))))
