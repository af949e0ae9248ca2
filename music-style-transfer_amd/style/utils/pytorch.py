"""Tensor helpers with the reference's names and behaviour (style/utils/pytorch.py:7-94).

On the MI355X path the model does not execute these (the fused HIP kernels replace the
reshape / broadcast-concat glue), but whole-module snapshots written by train-model.py:156-160
pickle `style.utils.pytorch.Distributed` / `.LSTM` by class path, and user code may import the
functions, so they exist here with identical semantics.
"""
import numpy as np
import torch
from torch import nn


def squash_dims(tensor, dim_begin, dim_end=None):
    """Merge dims [dim_begin, dim_end) into one (negative dim_begin counts from the end)."""
    nd = tensor.dim()
    if dim_end is None:
        dim_end = nd
    if dim_begin < 0:
        dim_begin, dim_end = dim_begin + nd, dim_end + nd
    shape = list(tensor.shape)
    merged = int(np.prod(shape[dim_begin:dim_end]))
    return tensor.view(*shape[:dim_begin], merged, *shape[dim_end:])


class LSTM(nn.LSTM):
    """nn.LSTM whose final states are batch-first too when batch_first=True."""

    def forward(self, *args, **kwargs):
        out, (h, c) = super().forward(*args, **kwargs)
        if self.batch_first:
            h, c = h.transpose(0, 1), c.transpose(0, 1)
        return out, (h, c)


class Distributed(nn.Module):
    """Apply `module` over the first depth+1 dims flattened into one batch dim."""

    def __init__(self, module, depth=1):
        super().__init__()
        self.module = module
        self.depth = depth

    def forward(self, x):
        head = tuple(x.shape[:self.depth + 1])
        y = self.module(x.reshape(-1, *x.shape[self.depth + 1:]))
        return self.view_tuple(y, *head)

    def __repr__(self):
        return f'{self.__class__.__name__} ({self.module!r})'

    @classmethod
    def view_tuple(cls, x, *head):
        if isinstance(x, tuple):
            return tuple(cls.view_tuple(t, *head) for t in x)
        return x.view(*head, *x.shape[1:])


def cat_with_broadcast(tensors, dim=0):
    """torch.cat after expanding every tensor to the common shape on the other dims."""
    assert len(tensors) and all(t.dim() == tensors[0].dim() for t in tensors)
    target = [max(t.shape[d] for t in tensors) for d in range(tensors[0].dim())]
    out = []
    for t in tensors:
        shape = list(target)
        shape[dim] = t.shape[dim]
        out.append(t.expand(*shape))
    return torch.cat(out, dim=dim)


def safe_sqrt(x):
    if x == 0:
        return torch.tensor(0., requires_grad=x.requires_grad) * x
    return torch.sqrt(x)


def get_mean(tensors, weights=None, mean_type='arithmetic'):
    n = len(tensors)
    if weights is None:
        weights = np.ones(n) / n
    if mean_type == 'arithmetic':
        return sum(w * t for t, w in zip(tensors, weights))
    if mean_type == 'harmonic':
        return 1 / get_mean([1 / t for t in tensors], weights=weights)
    if mean_type == 'geometric':
        return torch.stack(tensors).prod(0) ** (1 / n)
    if mean_type == 'quadratic':
        return safe_sqrt(get_mean([t ** 2 for t in tensors], weights=weights))
    raise ValueError(f'Unsupported mean type: {mean_type}')
