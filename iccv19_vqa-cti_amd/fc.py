"""FCNet -- drop-in for the reference's src/fc.py:10-34 (same constructor, forward, state_dict keys
`main.<i>.{bias,weight_g,weight_v}` and the same RNG consumption at construction, so the same seed gives the same
parameters as the reference module).

Forward: each `[Dropout] -> weight_norm(Linear, dim=None) -> [act]` group is ONE launch of the MFMA GEMM with the
weight-norm factor g/||V||_F (a scalar, computed on the device by cti_wn_scale), the bias and the ReLU applied
in its epilogue; the normalised weight is never materialised (the reference re-materialises it every forward
through the weight_norm pre-forward hook)."""
import math
import weakref

import torch
import torch.nn as nn

from . import ops
from . import autograd as AG


_NO_BATCH = __import__("os").environ.get("CTI_NO_BATCHED_SCALES", "0") == "1"      # A/B knob: every layer refreshes its own scale
_scale_users = weakref.WeakSet()       # WNLinear layers that have asked for their scale: refreshed TOGETHER, in one launch pair, when stale


def refresh_stale_scales(device):
    """Refresh, on the CURRENT stream, the cached g / ||V||_F of every layer that has ever asked for it and whose parameters changed since: what
    ops.run_concurrently does before it forks (a layer's first asker refreshes ALL stale layers on its own stream -- see WNLinear.scale)."""
    stale = [m for m in _scale_users if m.weight_v.device == device and getattr(m, "_scale_key", None) != m._scale_key_now()]
    if stale:
        stale[0].scale()


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

    def apply(self, fn):
        """nn.Module.apply, then ops.invalidate_caches(): the usual passenger of .apply is an initialiser that writes through `param.data` (the reference's own
        utils.weights_init, src/utils.py:61-70,77: `m.weight.data.normal_(0.0, 0.02)`), which moves neither the autograd version nor the storage the derived
        weight caches are keyed on.  Every network of this package holds WNLinear layers, so any .apply over a tree that contains one refreshes the caches."""
        r = super().apply(fn)
        ops.invalidate_caches()
        return r

    def scale(self):
        """g / ||V||_F on the device, cached until weight_v / weight_g change (optimizer steps bump `_version`, .to() / load_state_dict
        change the storage): in eval mode the norm of a 6M-element weight is computed once, not per forward."""
        key = self._scale_key_now()
        if getattr(self, "_scale_key", None) != key:
            # A training step makes every layer's scale stale at once; the first layer to notice refreshes all the layers that have ever asked
            # (same device) in ONE batched call instead of one launch pair per layer.  The refresh runs on the CURRENT stream: layers whose
            # scale is consumed on another stream must be forked from it after this point (the models' aux-stream sections use no WNLinear).
            _scale_users.add(self)
            dev = self.weight_v.device
            stale = [m for m in _scale_users if m.weight_v.device == dev and getattr(m, "_scale_key", None) != m._scale_key_now()]
            if self not in stale:
                stale.append(self)
            if len(stale) == 1 or _NO_BATCH:
                self._scale_val = ops.wn_scale(self.weight_v.detach(), self.weight_g.detach())
                self._scale_key = key
            else:
                vals = ops.wn_scale_many([(m.weight_v.detach(), m.weight_g.detach()) for m in stale])
                for i, m in enumerate(stale):
                    m._scale_val = vals[i:i + 1]
                    m._scale_key = m._scale_key_now()
        return self._scale_val

    def _scale_key_now(self):
        return (self.weight_v.data_ptr(), self.weight_v._version, self.weight_g.data_ptr(), self.weight_g._version, ops._param_epoch[0])

    def planes(self):
        """weight_v as resident bf16 hi/lo operand planes (None in the exact-fp32 mode), cached like scale(): inference splits a weight once,
        not once per forward."""
        key = (self.weight_v.data_ptr(), self.weight_v._version, ops._param_epoch[0], ops.get_precision())
        if getattr(self, "_planes_key", None) != key:
            self._planes_val = ops.split_operand(self.weight_v.detach())
            self._planes_key = key
        return self._planes_val

    def forward(self, x, relu=False):
        if x.dtype == torch.bfloat16:
            # (round 5) bf16 activations of the plain-bf16 mode: the row-major bf16 matrix IS the product's A operand (no split pass), the output leaves as bf16
            # rows = the next consumer's operand (cti_gemm_bf16_rows).  Anything else widens first.
            if (_bf16_rows_ok(x, self.in_features, self.out_features) and not (torch.is_grad_enabled() and (x.requires_grad or self.weight_v.requires_grad
                                                                                            or self.weight_g.requires_grad or self.bias.requires_grad))):
                x2 = x.reshape(-1, x.shape[-1])
                y = ops.gemm_bf16_rows(x2, self.planes(), self.out_features, out_dtype=torch.bfloat16, scale=self.scale(), scale_div=self.out_features, bias=self.bias,
                                       relu=bool(relu))
                return y.view(x.shape[:-1] + (self.out_features,))
            x = ops.widen_bf16(x)
        if torch.is_grad_enabled() and (x.requires_grad or self.weight_v.requires_grad or self.weight_g.requires_grad or self.bias.requires_grad):
            return AG.WNLinearFn.apply(x, self.weight_v, self.weight_g, self.bias, relu, 1, self.scale())
        return ops.wn_linear(x, self.weight_v, self.scale(), self.out_features, self.bias, relu, w_planes=self.planes())


def _bf16_rows_ok(x, in_features, out_features=None):
    """The bf16-rows GEMM takes this input -- the same conditions as csrc/cti_gemm16.hip's gemm16_eligible / cti_gemm_bf16_rows (the laxer test sent a
    layer the kernel refuses into CTI_E_ALIGN instead of the widen-first path): plain-bf16 mode, K a multiple of 32 and at least four 32-deep stages (bias + scale
    ride behind a tile's first stage), output rows of a multiple of 4 elements, 16-B aligned input rows at a 16-B aligned origin, and at least a few 256-row
    tiles' worth of rows (its 256 x 256 tile)."""
    return (ops.get_precision() == "bf16" and x.is_cuda and in_features % 32 == 0 and in_features >= 128 and (out_features is None or out_features % 4 == 0)
            and x.numel() // max(1, x.shape[-1]) >= 1024 and x.stride(-1) == 1 and x.data_ptr() % 16 == 0 and (x.dim() < 2 or (x.stride(-2) * 2) % 16 == 0))


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


class HoistedProjection:
    """Model-level fusion of SURVEY.md 8f row N1: several single-layer FCNets `[Dropout, WNLinear(in, out), ReLU]` that read the SAME
    input (the `v_net` / `v_tucker` of every glimpse's pooling network read `v`, src/FFOE/base_model.py:57,130) run as ONE batched
    GEMM -- the input is split into bf16 planes once instead of once per network, and the launch has n times the tiles.
    Inference only (in train mode every network draws its own input-dropout mask, and autograd needs the per-network graph):
    `maybe(...)` returns None whenever the fusion does not apply and the caller takes the per-network path.
    The concatenated weights are cached until a parameter changes (`_version` / `data_ptr`)."""

    def __init__(self, nets, padded=()):
        """padded: further networks of the same form and input width whose output is NARROWER than the common one (TriAttention's v_tucker, 512
        wide, beside the pooling networks' 1024): they join the batched GEMM with zero rows up to the common width -- after `maybe` their
        outputs are in `last_padded` (full width: the caller reads the first out_features columns, row stride = the common width; empty when
        such a network does not have the batched form)."""
        self.nets = list(nets)
        self.padded = list(padded)
        self.last_padded = []
        self._key = None

    @staticmethod
    def _layer_of(n):
        mods = [m for m in n.main if not isinstance(m, nn.Dropout)]
        return mods[0] if len(mods) == 2 and isinstance(mods[0], WNLinear) and isinstance(mods[1], nn.ReLU) else None

    def _layers(self):
        main = [self._layer_of(n) for n in self.nets]
        if any(l is None for l in main) or len({(l.in_features, l.out_features) for l in main}) != 1:
            return None
        extra = [self._layer_of(n) for n in self.padded]
        if any(l is None or l.in_features != main[0].in_features or l.out_features > main[0].out_features or n.training for l, n in zip(extra, self.padded)):
            extra = []                                   # a narrower network of another form (e.g. act='Tanh'): the main ones still batch
        return main + extra

    def maybe(self, x, use_padded=True, relu=True):
        """x (..., in) -> list of (..., out) tensors, one per network, or None.  use_padded = False: the narrower networks stay out of the batch
        (their consumer will not take the hoisted output on this call: the extra rows would be computed and discarded).
        relu = False: the PRE-activation outputs scale * W x + bias (the hoisted glimpse loops add the residual's projection before the ReLU)."""
        self.last_padded = []
        if len(self.nets) + (len(self.padded) if use_padded else 0) < 2 or torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for n in self.nets + self.padded for p in n.parameters())):
            return None
        if any(n.training for n in self.nets):
            return None
        layers = self._layers()
        if layers is None:
            return None
        if not use_padded:
            layers = layers[:len(self.nets)]
        key = (ops._param_epoch[0], ops.get_precision(), bool(use_padded)) + tuple((p.data_ptr(), p._version) for l in layers for p in (l.weight_v, l.weight_g, l.bias))
        if key != self._key:
            with torch.no_grad():
                od = layers[0].out_features

                def rows(t, l):                       # a narrower layer: zero rows up to the common width
                    return t if l.out_features == od else torch.cat([t, t.new_zeros((od - l.out_features,) + tuple(t.shape[1:]))], 0)
                self._w = torch.cat([rows(l.weight_v.detach(), l) for l in layers], 0).contiguous()
                self._b = torch.cat([rows(l.bias.detach(), l) for l in layers], 0).contiguous()
                self._s = torch.cat([l.scale().view(1) for l in layers], 0).contiguous()
                self._wp = ops.split_operand(self._w)
            self._key = key
        n, out_dim = len(layers), layers[0].out_features
        x2 = x.reshape(-1, x.shape[-1])
        if x.dtype == torch.bfloat16:
            if _bf16_rows_ok(x2, layers[0].in_features):
                # (round 5) bf16 rows in -> bf16 rows out: no split pass over x, half the bytes written here and read by the pools / the attention
                y = ops.gemm_bf16_rows(x2, self._wp, n * out_dim, nb1=n, rA1=0, rB1=out_dim, M=x2.shape[0], N=out_dim, out_dtype=torch.bfloat16, scale=self._s,
                                       scale_div=out_dim, scale_bs=1, bias=self._b, bias_bs=out_dim, relu=bool(relu))
                outs = [y[i].view(x.shape[:-1] + (out_dim,)) for i in range(n)]
                self.last_padded = outs[len(self.nets):]
                return outs[:len(self.nets)]
            x2 = ops.widen_bf16(x2)
        y = ops.gemm_nt(x2, self._w, nb1=n, rA1=0, rB1=out_dim, M=x2.shape[0], N=out_dim, scale=self._s, scale_div=out_dim, scale_bs=1,
                        bias=self._b, bias_bs=out_dim, relu=bool(relu), B_planes=self._wp)
        outs = [y[i].view(x.shape[:-1] + (out_dim,)) for i in range(n)]
        self.last_padded = outs[len(self.nets):]      # the narrower networks' outputs (full width; the first out_features columns count), possibly none
        return outs[:len(self.nets)]


class BatchedLinears:
    """n WNLinear layers of one (in, out) shape as ONE batched GEMM (inference): on a shared input (`shared`: the residual projections q_prj[g] / a_prj[g] of one
    pooled vector) or on n stacked inputs (`stacked`: the shift projections of the hoisted glimpse loops' two sequences).  Weights, scales and biases are
    concatenated once and cached until a parameter changes, like HoistedProjection."""

    def __init__(self, layers):
        self.layers = list(layers)
        self._key = None

    def _refresh(self):
        ls = self.layers
        key = (ops._param_epoch[0], ops.get_precision()) + tuple((p.data_ptr(), p._version) for l in ls for p in (l.weight_v, l.weight_g, l.bias))
        if key != self._key:
            with torch.no_grad():
                self._w = torch.cat([l.weight_v.detach() for l in ls], 0).contiguous()
                self._b = torch.cat([l.bias.detach() for l in ls], 0).contiguous()
                self._s = torch.cat([l.scale().view(1) for l in ls], 0).contiguous()
                self._wp = ops.split_operand(self._w)
            self._key = key
        return len(ls), ls[0].out_features

    def shared(self, x, relu=False):
        """x (rows, in) -> (n, rows, out) = act(scale_i W_i x + bias_i)."""
        n, od = self._refresh()
        return ops.gemm_nt(x, self._w, nb1=n, rA1=0, rB1=od, M=x.shape[0], N=od, scale=self._s, scale_div=od, scale_bs=1, bias=self._b, bias_bs=od, relu=bool(relu),
                           B_planes=self._wp)

    def stacked(self, x, bias=True, relu=False):
        """x (n, rows, in) contiguous -> (n, rows, out) = act(scale_i W_i x_i [+ bias_i])."""
        n, od = self._refresh()
        rows = x.shape[1]
        return ops.gemm_nt(x.reshape(n * rows, x.shape[2]), self._w, nb1=n, rA1=rows, rB1=od, M=rows, N=od, scale=self._s, scale_div=od, scale_bs=1,
                           bias=self._b if bias else None, bias_bs=od if bias else 0, relu=bool(relu), B_planes=self._wp)
