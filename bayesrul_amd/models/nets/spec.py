"""Static layer tables of the two networks the reference defines for this path
(bayesrul/models/nets/inception.py:142-217, bayesrul/models/nets/linear.py:10-72); the
host-side mirror of the tables compiled into the native plan (csrc/plan.hip)."""
from __future__ import annotations

from typing import List, Tuple

# (layer name, is_conv, cout, cin, kernel)
def net_layers(net: str, win_length: int = 30, n_features: int = 18) -> List[Tuple[str, bool, int, int, int]]:
    if net == "inception":
        return [
            ("layers.0.conv1.0", True, 27, n_features, 1),
            ("layers.0.conv3.0", True, 27, n_features, 3),
            ("layers.0.conv5.0", True, 27, n_features, 5),
            ("layers.0.convpool.1", True, 27, n_features, 3),
            ("layers.1.branch1.0", True, 16, 108, 1),
            ("layers.1.branch2.0", True, 64, 108, 1),
            ("layers.1.branch2.2", True, 16, 64, 3),
            ("layers.1.branch3.0", True, 64, 108, 1),
            ("layers.1.branch3.2", True, 16, 64, 5),
            ("layers.1.branch4.1", True, 32, 108, 1),
            ("layers.3", False, 64, 80 * win_length, 0),
            ("last", False, 2, 64, 0),
        ]
    if net == "linear":
        return [
            ("layers.1", False, 256, win_length * n_features, 0),
            ("layers.3", False, 128, 256, 0),
            ("layers.5", False, 128, 128, 0),
            ("layers.7", False, 32, 128, 0),
            ("last", False, 2, 32, 0),
        ]
    raise ValueError(f"unknown net {net!r}")


def site_shapes(net: str, win_length: int = 30, n_features: int = 18):
    """`named_parameters` order: weight then bias per layer."""
    out = []
    for name, conv, cout, cin, k in net_layers(net, win_length, n_features):
        out.append((name + ".weight", (cout, cin, k) if conv else (cout, cin)))
        out.append((name + ".bias", (cout,)))
    return out
