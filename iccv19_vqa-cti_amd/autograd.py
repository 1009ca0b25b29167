"""torch.autograd.Function wrappers: forward AND backward of every differentiable op of the CTI path run in the HIP library
(the reference relies on torch.autograd through src/fc.py, src/tc.py, src/bc.py, src/attention.py; nothing here falls back to
eager PyTorch arithmetic)."""
import torch

from . import ops


class WNLinearFn(torch.autograd.Function):
    """y = act(scale[n // div] * x @ V^T + bias), scale[i] = g[i] / ||V_i||_F with n_mats stacked sub-matrices V_i
    (n_mats = 1: one FCNet layer, src/fc.py:22-29; n_mats = R: the packed rank nets of src/tc.py:29-31)."""

    @staticmethod
    def forward(ctx, x, weight_v, weight_g, bias, relu, n_mats, scale=None):
        out_dim = weight_v.shape[0]
        if scale is None:                                        # a layer's cached / batched scale (WNLinear.scale()) or computed here
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
            dx = ops.gemm_nn(dzs, V.contiguous()).view(x.shape)                     # no transposed copy of the weight
        G = ops.gemm_tn(dzs, x.contiguous().view(-1, K))                           # (N, K) = dzs^T x
        dV, dg = ops.wn_bwd(G, V, g, ctx.n_mats)
        return dx, dV.view_as(V), dg.view_as(g), db, None, None, None


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


class TeffFn(torch.autograd.Function):
    """T_g (1,R,hr,hr,hr,G,1) -> T_eff (R,hr,hr,hr,G): the index scramble of src/Tensor.py:6-8; backward = inverse scramble."""

    @staticmethod
    def forward(ctx, T_g):
        ctx.shape = T_g.shape
        return ops.teff_scramble(T_g[0, :, :, :, :, :, 0])

    @staticmethod
    def backward(ctx, dTeff):
        return ops.teff_scramble(dTeff.contiguous(), inverse=True).view(ctx.shape)


class MBuildFn(torch.autograd.Function):
    """Modes 1+2 of Tensor.ModeProduct for all ranks (src/Tensor.py:6-13)."""

    @staticmethod
    def forward(ctx, Vr, Qr, Teff):
        ctx.save_for_backward(Vr, Qr, Teff)
        return ops.paralind_mbuild(Vr, Qr, Teff)

    @staticmethod
    def backward(ctx, dM):
        Vr, Qr, Teff = ctx.saved_tensors
        return ops.paralind_mbuild_bwd(dM, Vr, Qr, Teff)


class CoreFn(torch.autograd.Function):
    """Mode 3 + rank sum (src/Tensor.py:16-20, src/tc.py:50)."""

    @staticmethod
    def forward(ctx, M, Ar):
        ctx.save_for_backward(M, Ar)
        return ops.paralind_core(M, Ar)

    @staticmethod
    def backward(ctx, dout):
        M, Ar = ctx.saved_tensors
        return ops.paralind_core_bwd(dout, M, Ar)


class MBuildCoreFn(torch.autograd.Function):
    """MBuildFn + CoreFn in the MFMA modes with M kept ONLY as the bf16 hi/lo operand planes of the mode-3 GEMM (what the fused eval forward
    does): no fp32 M, no split pass over it; the backward rebuilds M = hi + lo for dAr and never needs it for the M-build gradients."""

    @staticmethod
    def supported(Vr, Ar, Teff):
        R, I, J, K, G = Teff.shape
        return (ops.get_precision() in ("bf16x3", "bf16", "f16f6") and I == J == K and (R * K) % 32 == 0 and Ar.shape[1] <= 8 and Vr.shape[0] > 0
                and Vr.shape[0] <= 65535)

    @staticmethod
    def forward(ctx, Vr, Qr, Teff, Ar):
        Mh, Ml = ops.paralind_mbuild_planes(Vr, Qr, Teff)
        B, V, Q, G = Vr.shape[0], Vr.shape[1], Qr.shape[1], Teff.shape[4]
        ctx.save_for_backward(Vr, Qr, Teff, Ar, Mh, Ml)
        ctx.G = G
        return ops.paralind_core_planes(Mh, Ml, Ar, B, V, Q, G)

    @staticmethod
    def backward(ctx, dout):
        Vr, Qr, Teff, Ar, Mh, Ml = ctx.saved_tensors
        dM, dAr = ops.paralind_core_bwd_planes(dout, Mh, Ml, Ar, ctx.G)
        dVr, dQr, dTeff = ops.paralind_mbuild_bwd(dM, Vr, Qr, Teff)
        return dVr, dQr, dTeff, dAr


class TriSoftmaxFn(torch.autograd.Function):
    """Masked softmax of TriAttention (src/attention.py:55-58).  The -inf fill happens on logits' DATA, untracked, exactly like the
    reference's `logits.data.masked_fill_`; masked positions have p = 0 and receive a zero gradient."""

    @staticmethod
    def forward(ctx, logits, mask):
        p = ops.masked_softmax_tri_(logits.detach(), mask)            # in place on the shared storage
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        return ops.masked_softmax_tri_bwd(p, dp), None


class BiSoftmaxFn(torch.autograd.Function):
    """Masked softmax of BiAttention (src/attention.py:35-39)."""

    @staticmethod
    def forward(ctx, logits, mask):
        p = ops.masked_softmax_bi_(logits.detach(), mask)
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        return ops.masked_softmax_bi_bwd(p, dp), None


class TriPoolFn(torch.autograd.Function):
    """einsum('bdv,bvqa,bdqi,bdaj->bdij') of src/tc.py:59."""

    @staticmethod
    def forward(ctx, vt, qt, at, w):
        ctx.save_for_backward(vt, qt, at, w)
        return ops.tri_pool(vt, qt, at, w)

    @staticmethod
    def backward(ctx, dout):
        vt, qt, at, w = ctx.saved_tensors
        dvt, dqt, dat, dw = ops.tri_pool_bwd(dout, vt, qt, at, w, ctx.needs_input_grad[3])
        return dvt, dqt, dat, dw


class BiPoolFn(torch.autograd.Function):
    """matmul pair + k-group sum-pool of src/bc.py:73-77 (w given) or the outer-product sum of src/bc.py:42-47 (w = None)."""

    @staticmethod
    def forward(ctx, vt, qt, w, k):
        ctx.save_for_backward(vt, qt, w)
        ctx.k = k
        return ops.bi_pool(vt, qt, w, k)

    @staticmethod
    def backward(ctx, dout):
        vt, qt, w = ctx.saved_tensors
        dvt, dqt, dw = ops.bi_pool_bwd(dout, vt, qt, w, ctx.k, w is not None and ctx.needs_input_grad[2])
        return dvt, dqt, dw, None


class BiLogitsFn(torch.autograd.Function):
    """logits[b,g,v,q] = s * sum_d vt h[g,d] qt + hb[g] (src/bc.py:52-58 / :63-68).  h_g is the weight-norm `g` of h (a 0-d
    tensor) or None when h is a plain parameter."""

    @staticmethod
    def forward(ctx, vt, qt, h, h_g, h_bias):
        G, D = h.shape[-3] if h.dim() == 4 else h.shape[0], h.shape[-1]
        h2 = h.reshape(G, D)
        scale = ops.wn_scale(h2.reshape(1, -1), h_g.reshape(1)) if h_g is not None else None
        ctx.save_for_backward(vt, qt, h, h_g if h_g is not None else vt.new_empty(0), scale if scale is not None else vt.new_empty(0))
        ctx.has_g = h_g is not None
        ctx.hb_shape = None if h_bias is None else h_bias.shape
        return ops.bi_logits(vt, qt, h2, scale, h_bias)

    @staticmethod
    def backward(ctx, dl):
        vt, qt, h, h_g, scale = ctx.saved_tensors
        G, D = (h.shape[-3] if h.dim() == 4 else h.shape[0]), h.shape[-1]
        h2 = h.reshape(G, D)
        dvt, dqt, Gh, dhb = ops.bi_logits_bwd(dl, vt, qt, h2, scale if ctx.has_g else None)
        if ctx.has_g:
            dh, dg = ops.wn_bwd(Gh, h2, h_g.reshape(1), 1)
            dg = dg.view(h_g.shape)
        else:
            dh, dg = Gh, None
        return dvt, dqt, dh.view(h.shape), dg, (None if ctx.hb_shape is None else dhb.view(ctx.hb_shape))


class RankPackFn(torch.autograd.Function):
    """The R rank nets' parameters packed for the batched kernels: (R*hr, h) weight_v, (R,) weight_g, (R*hr,) bias.  Same values as
    torch.cat / torch.stack; the backward hands each parameter its slice with three unbind calls instead of autograd's 3R narrow nodes
    (R = 32, three branches: 288 nodes per step on the host)."""

    @staticmethod
    def forward(ctx, R, *params):
        wv = torch.cat(params[:R], 0)
        g = torch.stack(params[R:2 * R])
        b = torch.cat(params[2 * R:], 0)
        ctx.R = R
        ctx.gshape = params[R].shape
        return wv, g, b

    @staticmethod
    def backward(ctx, dwv, dg, db):
        R = ctx.R
        out = [None]
        out += list(dwv.view(R, dwv.shape[0] // R, dwv.shape[1]).unbind(0)) if dwv is not None else [None] * R
        if dg is None:
            out += [None] * R
        else:
            gs = dg.unbind(0)
            out += list(gs) if gs[0].shape == ctx.gshape else [t.view(ctx.gshape) for t in gs]
        out += list(db.view(R, db.shape[0] // R).unbind(0)) if db is not None else [None] * R
        return tuple(out)


class RankNetsDropFn(torch.autograd.Function):
    """The R rank nets FCNet([h, hr]) of src/tc.py:29-31 in TRAIN mode.  Each net owns a Dropout on the SHARED input
    (src/fc.py:25-26), i.e. R independent masks: the input is expanded into R masked copies by one Philox kernel and the R
    projections run as ONE batched MFMA GEMM (per-batch weight-norm scale and bias); the backward is three batched GEMMs.
    Same distribution as the reference's 3 x R separate Dropout + Linear modules, ~40 launches instead of ~2000.
    For hr <= 16, h <= 512 (the real widths: 16 and 512) the masked copies are never materialised: the mask is drawn alone and the three
    contractions apply it to their operand fragments (cti_ranknets.hip)."""

    @staticmethod
    def forward(ctx, x, wv, g, b, relu, R, p):
        h = x.shape[-1]
        rows = x.numel() // h
        hr = wv.shape[0] // R
        x2 = x.contiguous().view(rows, h)
        scale = ops.wn_scale(wv.reshape(R, -1), g.reshape(-1))
        wvc = wv.contiguous()
        # fused route: only the (R, rows, h) byte mask exists; the three kernels of cti_ranknets.hip apply it to their operand fragments
        y = None
        if hr <= 16 and h % 4 == 0 and h <= 512 and rows > 0:
            mask = ops.dropout_mask((R, rows, h), p, x.device)
            y = ops.ranknets_drop_fwd(x2, mask, wvc, scale, b, R, p, relu)
        if y is not None:
            ctx.save_for_backward(x2, mask, y, wvc, g, scale)
            ctx.fused = True
        else:
            Xd, mask = ops.dropout(x2, p, copies=R)                                  # (R, rows, h)
            y = torch.empty((rows, R * hr), device=x.device, dtype=torch.float32)
            ops.gemm_nt(Xd.view(R * rows, h), wv, nb1=R, rA1=rows, rB1=hr, M=rows, N=hr, out=y, c_strides=(R * hr, 1), sC1=hr,
                        scale=scale, scale_div=max(hr, 1), scale_bs=1, bias=b, bias_bs=hr, relu=relu)
            ctx.save_for_backward(Xd, mask, y, wv, g, scale)
            ctx.fused = False
        ctx.cfg = (relu, R, p, hr, rows, h, x.shape)
        return y.view(x.shape[:-1] + (R * hr,))

    @staticmethod
    def backward(ctx, dy):
        Xd, mask, y, wv, g, scale = ctx.saved_tensors
        relu, R, p, hr, rows, h, xshape = ctx.cfg
        dzs, db = ops.act_bwd(dy, y, scale, hr, relu)                                # (rows, R*hr), (R*hr,)
        if ctx.fused:
            G = ops.ranknets_drop_dw(dzs, Xd, mask, R, p)                            # Xd is the plain (rows, h) input here
            dV, dg = ops.wn_bwd(G, wv, g, R)
            dx = ops.ranknets_drop_dx(dzs, wv, mask, R, p).view(xshape) if ctx.needs_input_grad[0] else None
            return dx, dV.view_as(wv), dg.view_as(g), db, None, None, None
        # dW_r = dzs_r^T @ Xd[r]  (contraction over the rows axis)
        dzsT = ops.transpose(dzs, rows, R * hr).view(R * hr, rows)
        XdT = ops.transpose(Xd, rows, h, R, rows * h)                                # (R, h, rows)
        G = ops.gemm_nt(dzsT, XdT.view(R * h, rows), nb1=R, rA1=hr, rB1=h, M=hr, N=h)
        dV, dg = ops.wn_bwd(G.view(R * hr, h), wv, g, R)
        dx = None
        if ctx.needs_input_grad[0]:
            # dXd[r] = dzs_r @ W_r, then back through the R masks and summed over the copies
            dzsR = dzs.view(rows, R, hr).permute(1, 0, 2).contiguous()               # layout change only
            WT = ops.transpose(wv.contiguous(), hr, h, R, hr * h)                    # (R, h, hr)
            dXd = ops.gemm_nt(dzsR.view(R * rows, hr), WT.view(R * h, hr), nb1=R, rA1=rows, rB1=h, M=rows, N=h)
            dXm = ops.dropout(dXd.view(R, rows, h), p, mask)
            dx = ops.sum_batches(dXm, R, rows * h).view(xshape)
        return dx, dV.view_as(wv), dg.view_as(g), db, None, None, None


# ---- rows either side of the CTI path ------------------------------------------------------------------------
class EmbeddingFn(torch.autograd.Function):
    """WordEmbedding lookup (src/language_model.py:40-44): table1 is the frozen copy concatenated when 'c' in op."""

    @staticmethod
    def forward(ctx, tokens, table0, table1, padding_idx):
        ctx.save_for_backward(tokens)
        ctx.shape, ctx.pad = tuple(table0.shape), padding_idx
        ctx.has1 = table1 is not None
        return ops.embedding(tokens, table0, table1)

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        rows, dim = ctx.shape
        d0 = ops.embedding_bwd(tokens, dout, 0, rows, dim, ctx.pad) if ctx.needs_input_grad[1] else None
        d1 = ops.embedding_bwd(tokens, dout, dim, rows, dim, ctx.pad) if (ctx.has1 and ctx.needs_input_grad[2]) else None
        return None, d0, d1, None


class GRUFn(torch.autograd.Function):
    """nn.GRU forward_all (src/language_model.py:91-96) with back-propagation through time in HIP kernels + MFMA GEMMs."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        out, save = ops.gru_forward(x, w_ih, w_hh, b_ih, b_hh, want_save=True)
        ctx.save_for_backward(x, w_ih, w_hh, save)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_ih, w_hh, save = ctx.saved_tensors
        dx, dW_ih, dW_hh, db_ih, db_hh = ops.gru_backward(dout, x, w_ih, w_hh, save, need_dx=ctx.needs_input_grad[0])
        return dx, dW_ih, dW_hh, db_ih, db_hh


class SwishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.swish(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.swish_bwd(x, dy)


class SeqSumFn(torch.autograd.Function):
    """(B,L,H) -> (B,H), the `.sum(1)` of src/FFOE/base_model.py:66,134."""

    @staticmethod
    def forward(ctx, x):
        ctx.L = x.shape[1]
        return ops.seq_sum(x)

    @staticmethod
    def backward(ctx, dout):
        return ops.seq_bcast_add(None, dout.contiguous(), ctx.L)


class SeqBcastAddFn(torch.autograd.Function):
    """x (B,L,H) + y (B,H)[:, None, :], the residual update of src/FFOE/base_model.py:61,131-132."""

    @staticmethod
    def forward(ctx, x, y):
        return ops.seq_bcast_add(x, y)

    @staticmethod
    def backward(ctx, dout):
        dy = ops.seq_sum(dout) if ctx.needs_input_grad[1] else None
        return (dout if ctx.needs_input_grad[0] else None), dy


class BCELogitsSumFn(torch.autograd.Function):
    """nn.BCEWithLogitsLoss(reduction='sum') (src/FFOE/train.py:28-33) -> 0-d tensor."""

    @staticmethod
    def forward(ctx, x, target):
        ctx.save_for_backward(x, target)
        rows = ops.bce_logits_sum(x, target)
        return ops.sum_batches(rows, rows.numel(), 1).view(())

    @staticmethod
    def backward(ctx, dl):
        x, target = ctx.saved_tensors
        return ops.bce_logits_bwd(x, target, dl.contiguous(), 1.0).view_as(x), None


class DistillationFn(torch.autograd.Function):
    """Distillation_Loss.forward (src/loss_function.py:21-24): mean_b KL(softmax(k/T) || softmax(x/T)) * alpha*T*T + BCE_sum/B * (1-alpha)."""

    @staticmethod
    def forward(ctx, x, knowledge, target, T, alpha):
        ctx.save_for_backward(x, knowledge, target)
        ctx.T, ctx.alpha = T, alpha
        B = x.shape[0]
        kl = ops.kd_rows(x, knowledge, T)
        bce = ops.bce_logits_sum(x, target)
        out = ops.sum_batches(kl, kl.numel(), 1, alpha=alpha * T * T / B)
        return ops.sum_batches(bce, bce.numel(), 1, alpha=(1.0 - alpha) / B, out=out, beta=1.0).view(())

    @staticmethod
    def backward(ctx, dl):
        x, knowledge, target = ctx.saved_tensors
        B = x.shape[0]
        dl = dl.contiguous()
        dx = ops.kd_rows_bwd(x, knowledge, dl, ctx.alpha * ctx.T * ctx.T / B, ctx.T)
        dx = ops.bce_logits_bwd(x, target, dl, (1.0 - ctx.alpha) / B, dx=dx, beta=1.0)
        return dx.view_as(x), None, None, None, None
