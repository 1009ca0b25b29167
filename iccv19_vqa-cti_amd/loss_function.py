"""Losses of the training step -- drop-in for the reference's src/loss_function.py:12-26 and the BCE criterion of
src/FFOE/train.py:28-33.  Forward and backward are row-reduction kernels of the HIP library."""
import torch.nn as nn

from . import autograd as AG


class BCEWithLogitsSum(nn.Module):
    """nn.BCEWithLogitsLoss(reduction='sum'); the trainer divides by the batch size (src/FFOE/trainer.py:189-190)."""

    def forward(self, input, target):
        return AG.BCELogitsSumFn.apply(input, target)


class Distillation_Loss(nn.Module):
    def __init__(self, T, alpha):
        super(Distillation_Loss, self).__init__()
        self.T = T
        self.alpha = alpha

    def forward(self, input, knowledge, target):
        return AG.DistillationFn.apply(input, knowledge, target, float(self.T), float(self.alpha))
