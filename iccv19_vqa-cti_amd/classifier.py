"""SimpleClassifier -- drop-in for the reference's src/classifier.py:11-28.

Layout kept for checkpoint compatibility: `main` is an nn.Sequential whose slots 0 and 3 are the two weight-normalised Linears
(state_dict keys `main.0.*`, `main.3.*`), slot 1 the activation, slot 2 the Dropout.  Arithmetic: both Linears are the path's MFMA GEMM
with the weight-norm scale and bias (and, for relu, the activation) in the epilogue -- split-K at batch-sized M; Swish and the
dropout are HIP kernels with their own backward."""
import torch.nn as nn

from . import autograd as AG
from .fc import WNLinear


class Swish(nn.Module):
    """x * sigmoid(x), the reference's alternative classifier activation (src/activation.py:17-22)."""

    def forward(self, x):
        return AG.SwishFn.apply(x)


_ACTIVATIONS = {"relu": nn.ReLU, "swish": Swish}


class SimpleClassifier(nn.Module):
    def __init__(self, in_dim, hid_dim, out_dim, args):
        super(SimpleClassifier, self).__init__()
        kind = getattr(args, "activation", None)
        if kind not in _ACTIVATIONS:
            raise AssertionError("%s is not supported yet!" % (kind,))
        # construction order = RNG order of the reference: first Linear, then the second
        hidden = WNLinear(in_dim, hid_dim)
        act = _ACTIVATIONS[kind]()
        drop = nn.Dropout(args.dropout)
        head = WNLinear(hid_dim, out_dim)
        self.main = nn.Sequential(hidden, act, drop, head)

    def forward(self, x):
        hidden, act, drop, head = self.main[0], self.main[1], self.main[2], self.main[3]
        fused_relu = isinstance(act, nn.ReLU)
        y = hidden(x, relu=fused_relu)
        if not fused_relu:
            y = act(y)
        y = AG.dropout(y, drop.p, self.training)
        return head(y)
