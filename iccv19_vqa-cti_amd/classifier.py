"""SimpleClassifier -- drop-in for the reference's src/classifier.py:11-28: weight-normalised
Linear -> relu | swish -> Dropout -> weight-normalised Linear, state_dict keys `main.0.*`, `main.3.*`.  Both Linears are the
MFMA GEMM with the weight-norm scale, bias (and ReLU) in the epilogue."""
import torch.nn as nn

from . import autograd as AG
from .fc import WNLinear


class Swish(nn.Module):
    """x * sigmoid(x) (src/activation.py:17-22)."""

    def forward(self, x):
        return AG.SwishFn.apply(x)


class SimpleClassifier(nn.Module):
    def __init__(self, in_dim, hid_dim, out_dim, args):
        super(SimpleClassifier, self).__init__()
        activation_dict = {'relu': nn.ReLU(), 'swish': Swish()}
        try:
            activation_func = activation_dict[args.activation]
        except Exception:
            raise AssertionError(str(getattr(args, 'activation', None)) + " is not supported yet!")
        layers = [
            WNLinear(in_dim, hid_dim),
            activation_func,
            nn.Dropout(args.dropout),
            WNLinear(hid_dim, out_dim),
        ]
        self.main = nn.Sequential(*layers)

    def forward(self, x):
        first, act, drop, last = self.main
        if isinstance(act, nn.ReLU):
            h = first(x, relu=True)
        else:
            h = act(first(x))
        h = AG.dropout(h, drop.p, self.training)
        return last(h)
