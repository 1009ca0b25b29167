"""BiAttention / TriAttention -- drop-ins for the reference's src/attention.py:14-59.

Both return `(p, logits)` with `logits` already -inf-filled on the all-zero rows of `v` (the reference fills them
in place, attention.py:37,56).  The zero-row mask is an exact bit test (every element +-0), the softmax is one
masked kernel (Bi: a wave per contiguous row; Tri: G interleaved softmaxes over the strided (v,q,a) axis)."""
import torch
import torch.nn as nn

from . import ops
from . import autograd as AG
from .bc import BCNet
from .tc import TCNet, _needs_grad


class BiAttention(nn.Module):
    def __init__(self, x_dim, y_dim, z_dim, glimpse, dropout=[.2, .5]):
        super(BiAttention, self).__init__()
        self.glimpse = glimpse
        self.logits = BCNet(x_dim, y_dim, z_dim, glimpse, dropout=dropout, k=3)._weight_norm_h_mat()

    def forward(self, v, q, v_mask=True):
        """
        v: [batch, k, vdim]
        q: [batch, qdim]
        """
        p, logits = self.forward_all(v, q, v_mask)
        return p, logits

    def forward_all(self, v, q, v_mask=True):
        return self._forward_all(v, q, v_mask)

    def _forward_all(self, v, q, v_mask=True, _v_projected=None, _mask=None):
        """forward_all; _v_projected / _mask (the model forwards' private arguments, eval only): logits.v_net(v) and zero_row_mask(v) already computed."""
        if not torch.is_grad_enabled() and not self.logits.training:
            # eval: logits, mask and softmax in ONE launch of the logits kernel (cti_biattention_fwd)
            fused = self.logits._attention(v, q, (ops.zero_row_mask(v) if _mask is None else _mask) if v_mask else None, v_projected=_v_projected)
            if fused is not None:
                return fused
        logits = self.logits(v, q)                                  # b x g x v x q
        mask = ops.zero_row_mask(v) if v_mask else None
        if _needs_grad(logits):
            p = AG.BiSoftmaxFn.apply(logits, mask)
        else:
            p = ops.masked_softmax_bi_(logits, mask)
        return p, logits


class TriAttention(nn.Module):
    def __init__(self, v_dim, q_dim, a_dim, h_dim, h_out, rank, glimpse, k, dropout=[.2, .5]):
        super(TriAttention, self).__init__()
        self.glimpse = glimpse
        self.TriAtt = TCNet(v_dim, q_dim, a_dim, h_dim, h_out, rank, glimpse, dropout=dropout, k=k)

    def forward(self, v, q, a, _v_tucked=None, _v_rep=1):
        """_v_tucked / _v_rep (the model forwards' private arguments): relu(TriAtt.v_tucker(v)) already computed in the batched projection
        of the glimpses' pooling networks, one block per image when the batch repeats every image _v_rep times."""
        t = self.TriAtt
        if self.glimpse >= 2 and t._fusable(v, q, a):
            # eval: logits, mask and softmax in ONE library call (cti_triattention_forward)
            tucker, rank = t._fused_args()
            return ops.triattention_forward(v if v.dtype == torch.bfloat16 else v.float(), q.float(), a.float(), tucker, rank, t.T_g.detach(), relu=(t._act == 'ReLU'), prepared=t._prep,
                                            v_tucked=_v_tucked if t._act == 'ReLU' else None, v_rep=_v_rep)
        logits, mask, partials = self.TriAtt(v, q, a, _want_mask=True, _want_sm_partials=True)
        if logits.dim() != 5:
            # glimpse == 1: TCNet.forward squeezed G away and the reference's mask expand (attention.py:55) raises
            raise RuntimeError("TriAttention needs glimpse >= 2 (the reference fails the same way: a 5-D mask is "
                               "expanded to the 4-D logits at src/attention.py:55)")
        if _needs_grad(logits):
            p = AG.TriSoftmaxFn.apply(logits, mask)
        elif partials is not None:                         # the mode-3 GEMM already reduced its outputs: the softmax reads the logits once
            p = ops.masked_softmax_tri_from_partials_(logits, mask, partials)
        else:
            p = ops.masked_softmax_tri_(logits, mask)
        return p, logits


class StackedAttention(nn.Module):
    """SAN baseline (src/attention.py:62-152) is a different model (`--model san`), outside the CTI hot path."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("StackedAttention is outside the CTI hot path; use the reference's class")
