"""gpytorch.utils.broadcasting._mul_broadcast_shape (used at models/nonstationary_models.py:104,107)."""
import torch


def _mul_broadcast_shape(*shapes, error_msg=None):
    try:
        return torch.broadcast_shapes(*shapes)
    except RuntimeError:
        raise RuntimeError(error_msg or 'Shapes are not broadcastable for mul operation')
