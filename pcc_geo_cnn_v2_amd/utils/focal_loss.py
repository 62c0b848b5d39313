"""Focal loss (training distortion term), /root/reference/src/utils/focal_loss.py:5-12, as one
deterministic HIP reduction (wavefront shuffles, fixed block order)."""
import torch

from .. import ops


def focal_loss(ctx, y_true, y_pred, gamma=2, alpha=0.9):
    return ops.focal_loss(ctx, y_true.contiguous().to(torch.float32), y_pred.contiguous().to(torch.float32),
                          float(gamma), float(alpha))
