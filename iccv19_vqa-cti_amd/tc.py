"""TCNet -- drop-in for the reference's src/tc.py (Compact Trilinear Interaction, PARALIND decomposition).

Same constructor / forward / forward_with_weights signatures, attributes and state_dict layout
(`T_g (1,R,hr,hr,hr,G,h_out)`, `{v,q,a}_tucker.main.1.*`, `{v,q,a}_net.<r>.main.1.*`).

forward (src/tc.py:41-52): 3 Tucker projections (one MFMA GEMM each), the 3 x R rank nets as 3 packed h -> R*hr
GEMMs with per-rank weight-norm scales in the epilogue, then T_eff scramble -> modes 1+2 (M build) -> mode 3 + rank
sum as one batched MFMA GEMM.  Inference (no autograd, eval mode) is ONE C-ABI call (cti_tcnet_forward: ~700 torch
launches of the reference become 24); under autograd the same kernels run op by op through autograd Functions whose
backward is HIP as well.
forward_with_weights (src/tc.py:54-61): 3 projections + one fused trilinear sum-pool kernel."""
import torch
import torch.nn as nn

from . import ops
from . import autograd as AG
from .fc import FCNet


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in ts)


class TCNet(nn.Module):

    def apply(self, fn):
        r = super().apply(fn)            # (an initialiser writing through .data moves no cache key: fc.WNLinear.apply)
        ops.invalidate_caches()
        return r
    def __init__(self, v_dim, q_dim, a_dim, h_dim, h_out, rank, glimpse, act='ReLU', dropout=[.2, .5], k=1):
        super(TCNet, self).__init__()
        self.v_dim = v_dim
        self.q_dim = q_dim
        self.a_dim = a_dim
        self.h_out = h_out
        self.rank = rank
        self.h_dim = h_dim * k
        self.hv_dim = int(h_dim / rank)
        self.hq_dim = int(h_dim / rank)
        self.ha_dim = int(h_dim / rank)
        self._act = act

        self.v_tucker = FCNet([v_dim, self.h_dim], act=act, dropout=dropout[1])
        self.q_tucker = FCNet([q_dim, self.h_dim], act=act, dropout=dropout[0])
        self.a_tucker = FCNet([a_dim, self.h_dim], act=act, dropout=dropout[0])
        if self.h_dim < 1024:
            self.a_tucker = FCNet([a_dim, self.h_dim], act=act, dropout=dropout[0])      # built twice, like tc.py:26,28
            self.v_net = nn.ModuleList([FCNet([self.h_dim, self.hv_dim], act=act, dropout=dropout[1]) for _ in range(rank)])
            self.q_net = nn.ModuleList([FCNet([self.h_dim, self.hq_dim], act=act, dropout=dropout[0]) for _ in range(rank)])
            self.a_net = nn.ModuleList([FCNet([self.h_dim, self.ha_dim], act=act, dropout=dropout[0]) for _ in range(rank)])
            if h_out > 1:
                self.ho_dim = int(h_out / rank)
                h_out = self.ho_dim
            self.T_g = nn.Parameter(torch.Tensor(1, rank, self.hv_dim, self.hq_dim, self.ha_dim, glimpse, h_out).normal_())
        self.dropout = nn.Dropout(dropout[1])                 # constructed and unused, like tc.py:38

    # ---- packed rank nets: R x FCNet([h, hr]) == one (R*hr, h) GEMM with per-rank scale ---------------------------
    @staticmethod
    def _last_linear(net):
        return net.main[-2] if isinstance(net.main[-1], nn.ReLU) else net.main[-1]

    def _rank_pack(self, nets):
        cache = self.__dict__.setdefault("_lins_cache", {})               # the module structure is static: resolve the R Linear layers once
        lins = cache.get(id(nets))
        if lins is None or len(lins) != len(nets):
            lins = cache[id(nets)] = [self._last_linear(n) for n in nets]
        if torch.is_grad_enabled() and any(l.weight_v.requires_grad or l.weight_g.requires_grad or l.bias.requires_grad for l in lins):
            return AG.RankPackFn.apply(len(lins), *[l.weight_v for l in lins], *[l.weight_g for l in lins], *[l.bias for l in lins])
        wv = torch.cat([l.weight_v for l in lins], 0)                      # (R*hr, h)
        g = torch.stack([l.weight_g for l in lins])                        # (R,)
        b = torch.cat([l.bias for l in lins], 0)                           # (R*hr,)
        return wv, g, b

    @staticmethod
    def _drop_p(net):
        return max([m.p for m in net.main if isinstance(m, nn.Dropout)] + [0.0])

    def _rank_proj(self, x, nets):
        if self._act not in ('ReLU', ''):
            # any other nn activation (the reference builds it by name, src/fc.py:24): the R rank nets run one by one through FCNet (GEMM on
            # the HIP library, the activation as its own module on the GEMM output) -- the packed single-GEMM forms below fuse only ReLU / none
            return torch.cat([n(x) for n in nets], dim=-1)
        relu = self._act == 'ReLU'
        wv, g, b = self._rank_pack(nets)
        if self.training and self._drop_p(nets[0]) > 0:
            # train mode: every rank net draws its OWN dropout mask on the shared input (src/fc.py:25-26 inside each of the R
            # FCNets): R masked copies of the input, one batched GEMM
            return AG.RankNetsDropFn.apply(x, wv, g, b, relu, len(nets), self._drop_p(nets[0]))
        if _needs_grad(x, wv, g, b):
            return AG.WNLinearFn.apply(x, wv, g, b, relu, len(nets))
        hr = wv.shape[0] // len(nets)
        scale = ops.wn_scale(wv.view(len(nets), -1), g)
        return ops.wn_linear(x, wv, scale, hr, b, relu)

    def _fused_args(self):
        """(tucker, rank) argument lists of ops.tcnet_forward; the packed rank weights are cached until a parameter
        changes (optimizer steps bump `_version`, load_state_dict / .to() change `data_ptr`)."""
        nets = (self.v_net, self.q_net, self.a_net)
        key = (ops._param_epoch[0],) + tuple((p.data_ptr(), p._version) for ns in nets for n in ns for p in n.parameters())
        if getattr(self, "_pack_key", None) != key:
            with torch.no_grad():
                self._pack = [tuple(t.detach() for t in self._rank_pack(ns)) for ns in nets]
            self._pack_key = key
        tucker = []
        for net in (self.v_tucker, self.q_tucker, self.a_tucker):
            l = self._last_linear(net)
            tucker.append((l.weight_v.detach(), l.weight_g.detach(), l.bias.detach()))
        # the batch-independent part of the fused forward (weight-norm scales, T_eff, the weights' operand planes), rebuilt only when a
        # parameter, T_g or the precision mode changes: inference holds its weights in GEMM-operand form
        pkey = (key, tuple((t.data_ptr(), t._version) for tk in tucker for t in tk), (self.T_g.data_ptr(), self.T_g._version), ops.get_precision())
        if getattr(self, "_prep_key", None) != pkey:
            self._prep = ops.tcnet_prepare(tucker, self._pack, self.T_g.detach())
            self._prep_key = pkey
        return tucker, self._pack

    def _fusable(self, *inputs):
        return (self._act in ('ReLU', '') and not self.training and not _needs_grad(*inputs, *self.parameters())
                and all(len([m for m in n.main if hasattr(m, "weight_v")]) == 1 for n in (self.v_tucker, self.q_tucker, self.a_tucker)))

    def forward(self, v, q, a, _want_mask=False, _want_sm_partials=False):
        """_want_mask / _want_sm_partials (TriAttention's private arguments): also return zero_row_mask(v) and, third, the softmax partials
        the fused forward can leave behind (None when this call did not produce them)."""
        if self._fusable(v, q, a):
            tucker, rank = self._fused_args()
            res = ops.tcnet_forward(v.float(), q.float(), a.float(), tucker, rank, self.T_g.detach(), relu=(self._act == 'ReLU'),
                                    want_mask=_want_mask, prepared=self._prep, want_sm_partials=_want_sm_partials)
            if _want_sm_partials:
                return res[0].squeeze(4), res[1], res[2]
            if _want_mask:
                return res[0].squeeze(4), res[1]
            return res.squeeze(4)
        v_tucker = self.v_tucker(v)
        q_tucker = self.q_tucker(q)
        a_tucker = self.a_tucker(a)
        Vr = self._rank_proj(v_tucker, self.v_net)            # (B,V,R*hr)
        Qr = self._rank_proj(q_tucker, self.q_net)
        Ar = self._rank_proj(a_tucker, self.a_net)
        T = self.T_g
        if T.size(6) != 1:
            raise RuntimeError("TCNet.forward: h_out must be 1 (src/Tensor.py:6 cannot view the core otherwise)")
        if _needs_grad(Vr, Qr, Ar, T):
            Teff = AG.TeffFn.apply(T)
            if AG.MBuildCoreFn.supported(Vr, Ar, Teff):
                f_emb = AG.MBuildCoreFn.apply(Vr, Qr, Teff, Ar)
            else:
                M = AG.MBuildFn.apply(Vr, Qr, Teff)
                f_emb = AG.CoreFn.apply(M, Ar)
        else:
            Teff = ops.teff_scramble(T.detach()[0, :, :, :, :, :, 0])
            M = ops.paralind_mbuild(Vr, Qr, Teff)
            f_emb = ops.paralind_core(M, Ar)                   # (B,V,Q,A,G)
        if _want_sm_partials:
            return f_emb.squeeze(4), ops.zero_row_mask(v), None
        if _want_mask:
            return f_emb.squeeze(4), ops.zero_row_mask(v)
        return f_emb.squeeze(4)

    def forward_with_weights(self, v, q, a, w):
        return self._pool_projected(self.v_tucker(v), q, a, w)

    def _pool_projected(self, v_, q, a, w, v_rep=1):
        """forward_with_weights given v_ = v_tucker(v), b x v x d (the model forwards compute it for all glimpses in one batched GEMM).
        v_rep > 1 (inference): v_ holds one block per IMAGE, shared by v_rep consecutive batch rows."""
        q_ = self.q_tucker(q)
        a_ = self.a_tucker(a)
        w = w.float()
        if v_.dtype == torch.bfloat16:
            v_ = ops.widen_bf16(v_)
        if _needs_grad(v_, q_, a_, w):
            return AG.TriPoolFn.apply(v_.repeat_interleave(v_rep, 0) if v_rep > 1 else v_, q_, a_, w)
        return ops.tri_pool(v_, q_, a_, w, v_rep=v_rep)
