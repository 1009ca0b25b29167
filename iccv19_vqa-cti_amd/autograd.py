"""torch.autograd.Function wrappers: forward AND backward of every differentiable op of the CTI path run in the HIP library
(the reference relies on torch.autograd through src/fc.py, src/tc.py, src/bc.py, src/attention.py; nothing here falls back to
eager PyTorch arithmetic)."""
import torch

from . import ops


class WNLinearFn(torch.autograd.Function):
    """y = act(scale[n // div] * x @ V^T + bias), scale[i] = g[i] / ||V_i||_F with n_mats stacked sub-matrices V_i
    (n_mats = 1: one FCNet layer, src/fc.py:22-29; n_mats = R: the packed rank nets of src/tc.py:29-31)."""

    @staticmethod
    def forward(ctx, x, weight_v, weight_g, bias, relu, n_mats):
        out_dim = weight_v.shape[0]
        scale = ops.wn_scale(weight_v.reshape(n_mats, -1), weight_g.reshape(-1))
        y = ops.wn_linear(x, weight_v, scale, out_dim // n_mats, bias, relu)
        ctx.save_for_backward(x, y, weight_v, weight_g, scale)
        ctx.relu, ctx.n_mats = relu, n_mats
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, V, g, scale = ctx.saved_tensors
        N, K = V.shape
        dzs, db = ops.act_bwd(dy, y, scale, N // ctx.n_mats, ctx.relu)            # (rows, N), (N,)
        dx = None
        if ctx.needs_input_grad[0]:
            Vt = ops.transpose(V.contiguous(), N, K).view(K, N)                    # (K, N): contraction axis contiguous
            dx = ops.gemm_nt(dzs, Vt).view(x.shape)
        G = ops.gemm_tn(dzs, x.contiguous().view(-1, K))                           # (N, K) = dzs^T x
        dV, dg = ops.wn_bwd(G, V, g, ctx.n_mats)
        return dx, dV.view_as(V), dg.view_as(g), db, None, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        y, mask = ops.dropout(x, p)
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.dropout(dy, ctx.p, mask), None


def dropout(x, p, training):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p)
