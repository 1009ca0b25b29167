"""BCNet -- drop-in for the reference's src/bc.py (bilinear connect network, Kim et al. BAN).

forward: the three h_out branches of src/bc.py:41-68; the `h_out <= 32` branch never materialises the
(B,G,V,D) broadcast product of bc.py:55 (0.9 GB at B=256, G=8): the bilinear logits kernel folds h into the v row
on the fly.  forward_with_weights (bc.py:70-78): two projections + one fused bilinear sum-pool (with the k-group
sum-pooling of bc.py:75-77 inside).  Under autograd the backward runs in HIP too (autograd.py)."""
import torch
import torch.nn as nn

from . import ops
from . import autograd as AG
from .fc import FCNet, WNLinear
from .tc import _needs_grad


class BCNet(nn.Module):
    """Simple class for non-linear bilinear connect network"""

    def apply(self, fn):
        r = super().apply(fn)            # (an initialiser writing through .data moves no cache key: fc.WNLinear.apply)
        ops.invalidate_caches()
        return r

    def __init__(self, v_dim, q_dim, h_dim, h_out, act='ReLU', dropout=[.2, .5], k=1):
        super(BCNet, self).__init__()
        self.c = 32
        self.k = k
        self.v_dim = v_dim; self.q_dim = q_dim
        self.h_dim = h_dim; self.h_out = h_out

        self.v_net = FCNet([v_dim, h_dim * self.k], act=act, dropout=dropout[0])
        self.q_net = FCNet([q_dim, h_dim * self.k], act=act, dropout=dropout[0])
        self.dropout = nn.Dropout(dropout[1])  # attention
        if 1 < k:
            self.p_net = nn.AvgPool1d(self.k, stride=self.k)

        if None == h_out:
            pass
        elif h_out <= self.c:
            self.h_mat = nn.Parameter(torch.Tensor(1, h_out, 1, h_dim * self.k).normal_())
            self.h_bias = nn.Parameter(torch.Tensor(1, h_out, 1, 1).normal_())
        else:
            self.h_net = WNLinear(h_dim * self.k, h_out)

    def _weight_norm_h_mat(self):
        """What `weight_norm(BCNet, name='h_mat', dim=None)` (src/attention.py:19-20) does to the parameters:
        h_mat -> h_mat_g () = ||h_mat||_F and h_mat_v = h_mat, registered after h_bias."""
        h = self.h_mat.data
        del self._parameters['h_mat']
        self.h_mat_g = nn.Parameter(torch.norm(h).clone())
        self.h_mat_v = nn.Parameter(h.clone())
        return self

    def _logits(self, v_, q_, h, h_g, h_bias):
        if _needs_grad(v_, q_, h, h_g, h_bias):
            return AG.BiLogitsFn.apply(v_, q_, h, h_g, h_bias)
        G, D = (h.shape[-3] if h.dim() == 4 else h.shape[0]), h.shape[-1]
        h2 = h.reshape(G, D)
        scale = ops.wn_scale(h2.reshape(1, -1), h_g.reshape(1)) if h_g is not None else None
        return ops.bi_logits(v_, q_, h2, scale, h_bias)

    def forward(self, v, q):
        if None == self.h_out:
            v_ = self.v_net(v)
            q_ = self.q_net(q)
            out = AG.BiPoolFn.apply(v_, q_, None, 1) if _needs_grad(v_, q_) else ops.bi_pool(v_, q_, None, 1)
            return out.unsqueeze(1)                                        # b x 1 x h_dim
        v_ = AG.dropout(self.v_net(v), self.dropout.p, self.training)     # bc.py:53 / :64
        q_ = self.q_net(q)
        if self.h_out <= self.c:
            if 'h_mat' in self._parameters:
                return self._logits(v_, q_, self.h_mat, None, self.h_bias)             # b x h_out x v x q
            return self._logits(v_, q_, self.h_mat_v, self.h_mat_g, self.h_bias)
        return self._logits(v_, q_, self.h_net.weight_v, self.h_net.weight_g, self.h_net.bias)

    def _attention(self, v, q, mask, v_projected=None):
        """BiAttention.forward_all on this network, eval / no-grad only: (p, logits) with the mask and the softmax applied in the logits kernel's own launch
        (ops.biattention_forward), or None when the caller must take forward() + the separate softmax (training, autograd, the pooled form)."""
        if self.h_out is None or self.training:
            return None
        if self.h_out <= self.c:
            h, h_g, h_bias = (self.h_mat, None, self.h_bias) if 'h_mat' in self._parameters else (self.h_mat_v, self.h_mat_g, self.h_bias)
        else:
            h, h_g, h_bias = self.h_net.weight_v, self.h_net.weight_g, self.h_net.bias
        if _needs_grad(v, q, h, h_g, h_bias) or torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return None
        v_ = self.v_net(v) if v_projected is None else v_projected      # (the model forward may have projected v on its auxiliary stream, beside the GRU)
        q_ = self.q_net(q)
        G, D = (h.shape[-3] if h.dim() == 4 else h.shape[0]), h.shape[-1]
        h2 = h.reshape(G, D)
        return ops.biattention_forward(v_, q_, h2, self._h_scale(h, h_g), h_bias, mask)

    def _h_scale(self, h, h_g):
        """g / ||h||_F of the weight-normed bilinear map (src/attention.py:19-20), cached like WNLinear.scale() until a parameter changes: on this inference-only
        path the two-launch norm stood on the dependent chain between the q projection and the logits of EVERY forward (23 us of BanModel's 1.09 ms)."""
        if h_g is None:
            return None
        key = (h.data_ptr(), h._version, h_g.data_ptr(), h_g._version, ops._param_epoch[0])
        if getattr(self, "_h_scale_key", None) != key:
            object.__setattr__(self, "_h_scale_val", ops.wn_scale(h.detach().reshape(1, -1), h_g.detach().reshape(1)))
            object.__setattr__(self, "_h_scale_key", key)
        return self._h_scale_val

    def forward_with_weights(self, v, q, w):
        return self._pool_projected(self.v_net(v), q, w)

    def _pool_projected(self, v_, q, w):
        """forward_with_weights given v_ = v_net(v) (the model forwards compute it for all glimpses in one batched GEMM)."""
        q_ = self.q_net(q)
        w = w.float()
        if v_.dtype == torch.bfloat16:
            v_ = ops.widen_bf16(v_)                                     # (the literal loop's pool reads fp32 rows; the hoisted loop's reads bf16)
        if _needs_grad(v_, q_, w):
            return AG.BiPoolFn.apply(v_, q_, w, self.k)
        return ops.bi_pool(v_, q_, w, self.k)
