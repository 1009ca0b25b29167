"""FCNet -- drop-in for the reference's src/fc.py:10-34 (same constructor, forward, state_dict keys
`main.<i>.{bias,weight_g,weight_v}` and the same RNG consumption at construction, so the same seed gives the same
parameters as the reference module).

Forward: each `[Dropout] -> weight_norm(Linear, dim=None) -> [act]` group is ONE launch of the MFMA GEMM with the
weight-norm factor g/||V||_F (a scalar, computed on the device by cti_wn_scale), the bias and the ReLU applied
in its epilogue; the normalised weight is never materialised (the reference re-materialises it every forward
through the weight_norm pre-forward hook)."""
import math

import torch
import torch.nn as nn

from . import ops
from . import autograd as AG


class WNLinear(nn.Module):
    """weight_norm(nn.Linear(in, out), dim=None) as parameters only: bias (out), weight_g (), weight_v (out, in),
    registered in the order torch's weight_norm leaves them (bias, weight_g, weight_v)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        lin = nn.Linear(in_features, out_features)              # reference initialiser and RNG stream (fc.py:22,27)
        self.bias = nn.Parameter(lin.bias.data.clone())
        self.weight_g = nn.Parameter(torch.norm(lin.weight.data).clone())    # dim=None: one Frobenius norm, shape ()
        self.weight_v = nn.Parameter(lin.weight.data.clone())

    def extra_repr(self):
        return "in_features=%d, out_features=%d, weight_norm(dim=None)" % (self.in_features, self.out_features)

    def scale(self):
        """g / ||V||_F on the device, cached until weight_v / weight_g change (optimizer steps bump `_version`, .to() / load_state_dict
        change the storage): in eval mode the norm of a 6M-element weight is computed once, not per forward."""
        key = (self.weight_v.data_ptr(), self.weight_v._version, self.weight_g.data_ptr(), self.weight_g._version)
        if getattr(self, "_scale_key", None) != key:
            self._scale_val = ops.wn_scale(self.weight_v.detach(), self.weight_g.detach())
            self._scale_key = key
        return self._scale_val

    def forward(self, x, relu=False):
        if torch.is_grad_enabled() and (x.requires_grad or self.weight_v.requires_grad or self.weight_g.requires_grad or self.bias.requires_grad):
            return AG.WNLinearFn.apply(x, self.weight_v, self.weight_g, self.bias, relu, 1)
        return ops.wn_linear(x, self.weight_v, self.scale(), self.out_features, self.bias, relu)


class FCNet(nn.Module):
    """Simple class for non-linear fully connect network (reference src/fc.py:10-34)."""

    def __init__(self, dims, act='ReLU', dropout=0):
        super(FCNet, self).__init__()
        layers = []
        for i in range(len(dims) - 2):
            if 0 < dropout:
                layers.append(nn.Dropout(dropout))
            layers.append(WNLinear(dims[i], dims[i + 1]))
            if '' != act:
                layers.append(getattr(nn, act)())
        if 0 < dropout:
            layers.append(nn.Dropout(dropout))
        layers.append(WNLinear(dims[-2], dims[-1]))
        if '' != act:
            layers.append(getattr(nn, act)())
        self.main = nn.Sequential(*layers)

    def forward(self, x):
        mods = list(self.main)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Dropout):
                x = AG.dropout(x, m.p, self.training)
                i += 1
            elif isinstance(m, WNLinear):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if isinstance(nxt, nn.ReLU):
                    x = m(x, relu=True)
                    i += 2
                else:
                    x = m(x, relu=False)
                    i += 1
            else:                      # an activation other than ReLU: its own torch module on the GEMM output
                x = m(x)
                i += 1
        return x
