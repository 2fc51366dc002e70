"""bayesrul/utils/miscellaneous.py:53-70 counterparts used by the hot path."""
import torch
import torch.nn as nn


def weights_init(m):
    """xavier-normal for conv weights, kaiming-normal for linear weights (biases untouched)."""
    if isinstance(m, (nn.Conv1d, nn.Conv2d)):
        torch.nn.init.xavier_normal_(m.weight)
    elif isinstance(m, nn.Linear):
        torch.nn.init.kaiming_normal_(m.weight)


def enable_dropout(model):
    for m in model.modules():
        if m.__class__.__name__.startswith("Dropout"):
            m.train()
