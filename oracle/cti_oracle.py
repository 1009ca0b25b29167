"""CPU ORACLE for the CTI hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the reference's algorithm for TCNet / BCNet / BiAttention /
TriAttention / FCNet / ModeProduct (aioz-ai/ICCV19_VQA-CTI).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`iccv19_vqa-cti_amd/`) never does and fails loudly
when its HIP library is missing.

Parity status: PINNED.  The reference has no tests of its own for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, captured in the build container by `tests/golden/make_golden.py` and
committed under `tests/golden/*.npz` (`tests/test_oracle_golden.py` checks every
one of them, including the Kolda-Bader known answer embedded at
reference src/Tensor.py:30-35).

Every function cites the reference lines it follows.  Parameters are passed as
a dict keyed exactly like the reference module's `state_dict()`.

All arithmetic runs in `dtype` (float32 by default, like the reference; float64
gives the "truth" used to separate reference rounding noise from kernel error).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "wn_scale", "wn_linear", "fcnet", "mode_product", "teff_from_tg", "teff_index_map",
    "tcnet_forward", "tcnet_forward_modeproduct", "tcnet_forward_with_weights", "tri_attention",
    "bcnet_forward", "bcnet_forward_with_weights", "bi_attention", "zero_row_mask", "norm_max_err",
]


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def _sub(params, prefix):
    """Sub-dict of `params` under `prefix` (with the prefix stripped)."""
    if not prefix:
        return params
    pl = len(prefix)
    return {k[pl:]: v for k, v in params.items() if k.startswith(prefix)}


def norm_max_err(x, ref):
    """max|x-ref| / max|ref| over entries where ref is finite (element-wise relative error is
    meaningless next to zeros and -inf).  Non-finite entries must match exactly."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    fin = np.isfinite(ref)
    if not np.array_equal(fin, np.isfinite(x)):
        return float("inf")
    if not np.array_equal(x[~fin], ref[~fin], equal_nan=True):
        return float("inf")
    if not fin.any():
        return 0.0
    den = np.max(np.abs(ref[fin]))
    return float(np.max(np.abs(x[fin] - ref[fin])) / (den if den > 0 else 1.0))


def wn_scale(weight_g, weight_v, dtype=np.float32):
    """weight_norm(..., dim=None): W = g * V / ||V||_F with ONE scalar g per layer (reference src/fc.py:22,27;
    torch.nn.utils.weight_norm `norm_except_dim(v, 2, -1)` = Frobenius norm of the whole tensor)."""
    v = np.asarray(weight_v, dtype=dtype)
    return np.asarray(weight_g, dtype=dtype) / np.sqrt(np.sum(v * v, dtype=dtype), dtype=dtype)


def wn_linear(x, weight_g, weight_v, bias, relu, dtype=np.float32):
    """One weight-normalised Linear (+ReLU): reference src/fc.py:22-24 / :27-29 (nn.Linear = x @ W.T + b)."""
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(weight_v, dtype=dtype) * wn_scale(weight_g, weight_v, dtype)
    y = x @ w.T + np.asarray(bias, dtype=dtype)
    return np.maximum(y, 0) if relu else y


def fcnet(x, params, prefix="", act="ReLU", dtype=np.float32):
    """FCNet.forward in eval mode (Dropout = identity): reference src/fc.py:13-34.  Layer indices follow the
    reference's nn.Sequential numbering (`main.<i>`), whatever mix of Dropout/Linear/act produced them."""
    p = _sub(params, prefix)
    idx = sorted({int(k.split(".")[1]) for k in p if k.startswith("main.") and k.endswith("weight_v")})
    assert act in ("ReLU", ""), "oracle restates the activations the hot path uses (ReLU / none)"
    for i in idx:
        x = wn_linear(x, p["main.%d.weight_g" % i], p["main.%d.weight_v" % i], p["main.%d.bias" % i],
                      relu=(act == "ReLU"), dtype=dtype)
    return x


def zero_row_mask(v):
    """mask[b, r] = (0 == v[b, r, :].abs().sum()): reference src/attention.py:36 and :55.
    For finite inputs this is exactly "every element of the row is +-0"."""
    return np.sum(np.abs(np.asarray(v)), axis=2) == 0


# ------------------------------------------------------------------------------------------------
# Tensor.ModeProduct -- literal restatement (views / transposes kept one for one)
# ------------------------------------------------------------------------------------------------
def mode_product(tensor, m1, m2, m3, dtype=np.float32):
    """Reference src/Tensor.py:3-20 (n_way = 3), restated step by step with numpy reshapes in place of torch
    `.view` (both reinterpret C-contiguous memory) and `np.ascontiguousarray` in place of `.contiguous()`.

    tensor: (1, I, J, K, G) or (1, I, J, K, G, 1);  m1 (B, V, I), m2 (B, Q, J), m3 (B, A, K).
    Returns the logical (B, V, Q, A, G) result as a C-contiguous array.
    """
    t = np.asarray(tensor, dtype=dtype)
    six = t.ndim == 6
    if six:                                     # TCNet passes T_g[:, r] = (1,I,J,K,G,h_out=1); .view below needs h_out==1
        assert t.shape[5] == 1
        t = t[..., 0]
    m1, m2, m3 = (np.asarray(m, dtype=dtype) for m in (m1, m2, m3))
    s0, s1, s2, s3, s4 = t.shape
    # mode-1 (Tensor.py:6-8)
    t1 = np.ascontiguousarray(np.swapaxes(t, 3, 2)).reshape(s0, s1, s2 * s3 * s4)
    tp = np.matmul(m1, t1)                                                   # (B, V, J*K*G) in (K,J,G) memory order
    t1 = np.swapaxes(tp.reshape(-1, tp.shape[1], s4, s3, s2), 4, 2)          # view(-1,V,G,K,J).transpose(4,2)
    # mode-2 (Tensor.py:11-13)
    u = np.swapaxes(np.swapaxes(t1, 2, 1), 4, 2)
    t2 = np.ascontiguousarray(u).reshape(-1, t1.shape[2], t1.shape[1] * t1.shape[3] * t1.shape[4])
    tp = np.matmul(m2, t2)
    t2 = tp.reshape(-1, tp.shape[1], t1.shape[4], t1.shape[3], t1.shape[1])
    t2 = np.swapaxes(np.swapaxes(t2, 4, 1), 4, 2)
    # mode-3 (Tensor.py:16-20)
    u = np.swapaxes(np.swapaxes(np.swapaxes(t2, 3, 1), 4, 2), 4, 3)
    t3 = np.ascontiguousarray(u).reshape(-1, t2.shape[3], t2.shape[2] * t2.shape[1] * t2.shape[4])
    tp = np.matmul(m3, t3)
    t3 = tp.reshape(-1, tp.shape[1], t2.shape[4], t2.shape[2], t2.shape[1])
    t3 = np.swapaxes(np.swapaxes(np.swapaxes(t3, 1, 4), 4, 2), 3, 2)
    return np.ascontiguousarray(t3)                       # 5-D (B,V,Q,A,G), as the reference returns it


def teff_index_map(hr, G):
    """Flat index into T (hr,hr,hr,G) of the element that acts at position (i,j,k,g): the scramble that
    src/Tensor.py:6-8 applies when G > 1 (`.view(-1, V, G, K, J)` re-interprets a (K,J,G)-ordered axis).
    Closed form: T_eff[i] = T[i].transpose(0,1).contiguous().view(G,hr,hr).permute(2,1,0)."""
    idx = np.arange(hr * hr * hr * G, dtype=np.int64).reshape(hr, hr, hr, G)
    out = np.empty_like(idx)
    for i in range(hr):
        out[i] = np.ascontiguousarray(np.swapaxes(idx[i], 0, 1)).reshape(G, hr, hr).transpose(2, 1, 0)
    return out


def teff_from_tg(T_g, dtype=np.float32):
    """T_g (1,R,hr,hr,hr,G,1) -> T_eff (R,hr,hr,hr,G) with
    ModeProduct(T_g[:, r], v_, q_, a_) == einsum('ijkg,bvi,bqj,bak->bvqag', T_eff[r], v_, q_, a_)."""
    T = np.asarray(T_g, dtype=dtype)
    assert T.ndim == 7 and T.shape[0] == 1 and T.shape[6] == 1
    R, hr, G = T.shape[1], T.shape[2], T.shape[5]
    assert T.shape[3] == hr and T.shape[4] == hr
    imap = teff_index_map(hr, G).reshape(-1)
    flat = T[0, :, :, :, :, :, 0].reshape(R, -1)
    return flat[:, imap].reshape(R, hr, hr, hr, G)


# ------------------------------------------------------------------------------------------------
# TCNet / TriAttention
# ------------------------------------------------------------------------------------------------
def _tc_dims(p):
    T = p["T_g"]
    return T.shape[1], T.shape[2], T.shape[5]            # R, hr, G


def _rank_proj(x_t, p, side, R, dtype):
    """x_t (B,N,h) -> (B,N,R,hr): the R independent FCNet([h, hr]) of src/tc.py:29-31,47-49."""
    outs = [fcnet(x_t, p, "%s_net.%d." % (side, r), dtype=dtype) for r in range(R)]
    return np.stack(outs, axis=2)


def tcnet_forward_modeproduct(v, q, a, params, prefix="", dtype=np.float32):
    """TCNet.forward exactly as written (src/tc.py:41-52): loop over ranks, literal ModeProduct, running sum."""
    p = _sub(params, prefix)
    R, hr, G = _tc_dims(p)
    vt = fcnet(v, p, "v_tucker.", dtype=dtype)
    qt = fcnet(q, p, "q_tucker.", dtype=dtype)
    at = fcnet(a, p, "a_tucker.", dtype=dtype)
    f = 0
    for r in range(R):
        v_ = fcnet(vt, p, "v_net.%d." % r, dtype=dtype)
        q_ = fcnet(qt, p, "q_net.%d." % r, dtype=dtype)
        a_ = fcnet(at, p, "a_net.%d." % r, dtype=dtype)
        f = mode_product(np.asarray(p["T_g"], dtype=dtype)[:, r], v_, q_, a_, dtype=dtype) + f
    return f[..., 0] if G == 1 else f                       # `.squeeze(4)` (tc.py:52) only bites when G == 1


def tcnet_forward(v, q, a, params, prefix="", dtype=np.float32):
    """TCNet.forward in closed form, contraction order = the reference's (v, then q, then a):
    out[b,v,q,a,g] = sum_r sum_ijk T_eff[r,i,j,k,g] V^[b,v,r,i] Q^[b,q,r,j] A^[b,a,r,k]."""
    p = _sub(params, prefix)
    R, hr, G = _tc_dims(p)
    vt = fcnet(v, p, "v_tucker.", dtype=dtype)
    qt = fcnet(q, p, "q_tucker.", dtype=dtype)
    at = fcnet(a, p, "a_tucker.", dtype=dtype)
    Vr = _rank_proj(vt, p, "v", R, dtype)
    Qr = _rank_proj(qt, p, "q", R, dtype)
    Ar = _rank_proj(at, p, "a", R, dtype)
    Te = teff_from_tg(p["T_g"], dtype)
    X = np.einsum("rijkg,bvri->bvrjkg", Te, Vr, optimize=True)
    M = np.einsum("bvrjkg,bqrj->bvqgrk", X, Qr, optimize=True)              # (B,V,Q,G,R,hr)
    B, V, Q = M.shape[:3]
    A = Ar.shape[1]
    Mm = M.reshape(B, V * Q * G, R * hr)
    out = np.matmul(Mm, np.swapaxes(Ar.reshape(B, A, R * hr), 1, 2))        # (B, VQG, A)
    out = out.reshape(B, V, Q, G, A).transpose(0, 1, 2, 4, 3)
    out = np.ascontiguousarray(out)
    return out[..., 0] if G == 1 else out


def tri_attention(v, q, a, params, prefix="", dtype=np.float32):
    """TriAttention.forward (src/attention.py:49-59): logits = TCNet.forward; rows of v that are all zero ->
    -inf over (q,a,g); softmax over the flattened (v,q,a) axis separately per (b,g).  Returns (p, logits)."""
    logits = tcnet_forward(v, q, a, params, prefix + "TriAtt.", dtype=dtype).copy()
    m = zero_row_mask(v)
    logits[m] = -np.inf
    B, V, Q, A, G = logits.shape
    x = logits.reshape(B, V * Q * A, G)
    with np.errstate(invalid="ignore"):
        mx = np.max(x, axis=1, keepdims=True)
        e = np.exp(x - mx)                                  # all-masked sample: (-inf) - (-inf) = nan, as in torch
        pr = e / np.sum(e, axis=1, keepdims=True, dtype=dtype)
    return pr.reshape(B, V, Q, A, G).astype(dtype), logits


def tcnet_forward_with_weights(v, q, a, w, params, prefix="", dtype=np.float32):
    """TCNet.forward_with_weights (src/tc.py:54-61): out[b,d] = sum_vqa v~[b,v,d] w[b,v,q,a] q~[b,q,d] a~[b,a,d]."""
    p = _sub(params, prefix)
    vt = fcnet(v, p, "v_tucker.", dtype=dtype)
    qt = fcnet(q, p, "q_tucker.", dtype=dtype)
    at = fcnet(a, p, "a_tucker.", dtype=dtype)
    return np.einsum("bvd,bvqa,bqd,bad->bd", vt, np.asarray(w, dtype=dtype), qt, at, optimize=True)


# ------------------------------------------------------------------------------------------------
# BCNet / BiAttention
# ------------------------------------------------------------------------------------------------
def bcnet_forward(v, q, params, prefix="", h_out=None, dtype=np.float32):
    """BCNet.forward, eval mode (src/bc.py:41-68).  Works on raw BCNet keys (`h_mat`) and on the
    weight-normalised form BiAttention creates (`h_mat_g`, `h_mat_v`; src/attention.py:19-20)."""
    p = _sub(params, prefix)
    vt = fcnet(v, p, "v_net.", dtype=dtype)
    qt = fcnet(q, p, "q_net.", dtype=dtype)
    if h_out is None:                                        # bc.py:42-47
        return np.einsum("bvd,bqd->bd", vt, qt, optimize=True)[:, None, :]
    if h_out <= 32:                                          # bc.py:52-58
        if "h_mat" in p:
            h = np.asarray(p["h_mat"], dtype=dtype)
        else:
            h = np.asarray(p["h_mat_v"], dtype=dtype) * wn_scale(p["h_mat_g"], p["h_mat_v"], dtype)
        H = h[0, :, 0, :]                                    # (G, d)
        logits = np.einsum("bvd,gd,bqd->bgvq", vt, H, qt, optimize=True)
        return logits + np.asarray(p["h_bias"], dtype=dtype)
    # bc.py:63-68: outer product -> weight-normalised Linear(h_dim*k, h_out)
    wn = np.asarray(p["h_net.weight_v"], dtype=dtype) * wn_scale(p["h_net.weight_g"], p["h_net.weight_v"], dtype)
    logits = np.einsum("od,bvd,bqd->bovq", wn, vt, qt, optimize=True)
    return logits + np.asarray(p["h_net.bias"], dtype=dtype)[None, :, None, None]


def bcnet_forward_with_weights(v, q, w, params, prefix="", k=1, dtype=np.float32):
    """BCNet.forward_with_weights (src/bc.py:70-78): out[b,d] = sum_vq v~[b,v,d] w[b,v,q] q~[b,q,d]; when k > 1,
    AvgPool1d(k, stride=k) * k = sum over consecutive groups of k channels."""
    p = _sub(params, prefix)
    vt = fcnet(v, p, "v_net.", dtype=dtype)
    qt = fcnet(q, p, "q_net.", dtype=dtype)
    out = np.einsum("bvd,bvq,bqd->bd", vt, np.asarray(w, dtype=dtype), qt, optimize=True)
    if k > 1:
        n = out.shape[1] // k
        out = out[:, :n * k].reshape(out.shape[0], n, k).sum(axis=2, dtype=dtype)
    return out


def bi_attention(v, q, params, prefix="", v_mask=True, dtype=np.float32):
    """BiAttention.forward_all (src/attention.py:30-40).  Returns (p (B,G,V,Q), logits with -inf filled)."""
    p = _sub(params, prefix)
    G = p["logits.h_bias"].shape[1]
    logits = bcnet_forward(v, q, p, "logits.", h_out=G, dtype=dtype).copy()
    if v_mask:
        m = zero_row_mask(v)                                  # (B,V)
        logits[np.broadcast_to(m[:, None, :, None], logits.shape)] = -np.inf
    B, G_, V, Q = logits.shape
    x = logits.reshape(B, G_, V * Q)
    with np.errstate(invalid="ignore"):
        mx = np.max(x, axis=2, keepdims=True)
        e = np.exp(x - mx)
        pr = e / np.sum(e, axis=2, keepdims=True, dtype=dtype)
    return pr.reshape(B, G_, V, Q).astype(dtype), logits
