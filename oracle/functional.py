"""Restatement of the four batched helpers the hot path uses (oracle; test infrastructure only).

Follows ``utils/functional.py`` of the reference:
  dot -> :14-16, t -> :19-21, mv -> :29-33, op -> :60-64.
Pinned against the reference itself by tests/golden/ref_functional.npz.
"""
import torch


def dot(v1, v2):
    """Batch dot product over the last dim (utils/functional.py:14-16)."""
    return (v1 * v2).sum(-1)


def t(x):
    """Matrix transpose of the two trailing dims (utils/functional.py:19-21)."""
    return x.transpose(-1, -2)


def mv(matrix, vector, invert=False):
    """matrix @ vector, or solve(matrix, vector) when invert (utils/functional.py:29-33)."""
    if not invert:
        return (matrix @ vector.unsqueeze(-1)).squeeze(-1)
    return torch.linalg.solve(matrix, vector.unsqueeze(-1)).squeeze(-1)


def op(v1, v2=None):
    """Outer product over the last dim (utils/functional.py:60-64)."""
    if v2 is None:
        v2 = v1
    return v1.unsqueeze(-1) @ v2.unsqueeze(-2)
