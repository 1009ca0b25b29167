"""Edge cases of the path on an MI355X against the oracle: degenerate sizes (one sample / object / token / candidate), feature widths
that are not multiples of the GEMM's K padding or tile sizes, ragged row counts around the tile boundaries, an empty batch, and
non-contiguous inputs.  Random-init modules; reference = the float64 oracle on the module's own state_dict."""
import numpy as np
import pytest
import torch

import cti_amd
from oracle import cti_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


@pytest.fixture(params=["fp32", "bf16x3"], autouse=True)
def precision(request):
    old = cti_amd.get_precision()
    cti_amd.set_precision(request.param)
    yield request.param
    cti_amd.set_precision(old)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def sd(m):
    return {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}


def check(x, ref, what, tol=TOL):
    x = x.detach().cpu().numpy()
    assert x.shape == ref.shape, (what, x.shape, ref.shape)
    if x.size:
        e = O.norm_max_err(x, ref)
        assert e < tol, "%s: %.3g" % (what, e)


SHAPES = [
    # B, V, Q, A, v_dim, q_dim, a_dim, h, R, G
    (1, 1, 1, 1, 8, 8, 8, 16, 4, 2),            # everything degenerate
    (2, 3, 2, 5, 33, 17, 9, 32, 4, 2),          # feature widths not multiples of 4 / 16 / 32
    (3, 36, 14, 4, 100, 60, 30, 64, 4, 3),      # odd glimpse count, K tails
    (5, 7, 3, 130, 24, 24, 20, 32, 2, 2),       # A just past a 128-column tile
    (2, 37, 14, 3, 40, 40, 40, 128, 32, 2),     # V*Q*G past the 1024 rows of one 256-row tile group
]


@pytest.mark.parametrize("shape", SHAPES)
def test_triattention_and_pool_shapes(shape):
    B, V, Q, A, vd, qd, ad, h, R, G = shape
    torch.manual_seed(sum(shape))
    att = cti_amd.TriAttention(vd, qd, ad, h, 1, R, G, 1).to(DEV).eval()
    net = cti_amd.TCNet(vd, qd, ad, h, 1, R, 1, k=2).to(DEV).eval()
    rs = np.random.RandomState(sum(shape))
    v = np.abs(rs.standard_normal((B, V, vd))).astype(np.float32)
    if V > 2:
        v[0, V - 1] = 0                                          # one masked object
    q = rs.standard_normal((B, Q, qd)).astype(np.float32)
    a = rs.standard_normal((B, A, ad)).astype(np.float32)
    with torch.no_grad():
        p, logits = att(T(v), T(q), T(a))
        pooled = net.forward_with_weights(T(v), T(q), T(a), p[..., 0])
    p_ref, l_ref = O.tri_attention(v, q, a, sd(att), dtype=np.float64)
    fin = np.isfinite(l_ref)
    assert np.array_equal(np.isfinite(logits.cpu().numpy()), fin)
    check(torch.where(torch.isfinite(logits), logits, torch.zeros_like(logits)), np.where(fin, l_ref, 0), "logits")
    check(p, p_ref, "p")
    check(pooled, O.tcnet_forward_with_weights(v, q, a, p_ref[..., 0], sd(net), dtype=np.float64), "pooled")


@pytest.mark.parametrize("B,V,Q,vd,qd,hd,G", [(1, 1, 1, 8, 8, 8, 1), (2, 5, 3, 33, 17, 20, 3), (3, 36, 14, 100, 60, 48, 8), (2, 36, 12, 64, 64, 256, 8)])
def test_biattention_and_pool_shapes(B, V, Q, vd, qd, hd, G):
    torch.manual_seed(B + V + Q + G)
    att = cti_amd.BiAttention(vd, qd, hd, G).to(DEV).eval()
    net = cti_amd.BCNet(vd, qd, hd, None, k=1).to(DEV).eval()
    rs = np.random.RandomState(B * 7 + V)
    v = np.abs(rs.standard_normal((B, V, vd))).astype(np.float32)
    if V > 2:
        v[0, 1] = 0
    q = rs.standard_normal((B, Q, qd)).astype(np.float32)
    with torch.no_grad():
        p, logits = att.forward_all(T(v), T(q))
        pooled = net.forward_with_weights(T(v), T(q), p[:, 0])
    p_ref, l_ref = O.bi_attention(v, q, sd(att), dtype=np.float64)
    fin = np.isfinite(l_ref)
    assert np.array_equal(np.isfinite(logits.cpu().numpy()), fin)
    check(torch.where(torch.isfinite(logits), logits, torch.zeros_like(logits)), np.where(fin, l_ref, 0), "logits", 1.5e-4)
    check(p, p_ref, "p", 1.5e-4)
    check(pooled, O.bcnet_forward_with_weights(v, q, p_ref[:, 0], sd(net), dtype=np.float64), "pooled")


@pytest.mark.parametrize("V,A", [(64, 3), (64, 6), (61, 3), (62, 3), (59, 6), (60, 6)])
@pytest.mark.parametrize("mode", ["bf16x3", "bf16", "f16f6"])
def test_fused_few_answer_kernel_at_its_lds_budget(V, A, mode):
    """hr = 16, glimpse 2, R = 32 and 59 ... 64 objects: the fused modes-1+2+3 kernel holds X (V*2*16*20 floats) + the sample's A^ block
    (A*512 floats) in LDS and refuses V >= 62 at A = 3, V >= 60 at A = 6.  The forward's plan must then carry M / A^ planes and take the
    M build + GEMM pair (round-2 ADVICE: the plan ignored the budget and the call raised CTI_E_UNSUPPORTED).  Both sides of the boundary."""
    old = cti_amd.get_precision()
    cti_amd.set_precision(mode)
    try:
        torch.manual_seed(V * 7 + A)
        att = cti_amd.TriAttention(48, 40, 24, 512, 1, 32, 2, 1).to(DEV).eval()
        rs = np.random.RandomState(V + A)
        v = np.abs(rs.standard_normal((2, V, 48))).astype(np.float32)
        v[1, V - 2:] = 0
        q = np.tanh(rs.standard_normal((2, 13, 40))).astype(np.float32)
        a = np.tanh(rs.standard_normal((2, A, 24))).astype(np.float32)
        with torch.no_grad():
            p, logits = att(T(v), T(q), T(a))
        p_ref, l_ref = O.tri_attention(v, q, a, sd(att), dtype=np.float64)
        fin = np.isfinite(l_ref)
        assert np.array_equal(np.isfinite(logits.cpu().numpy()), fin)
        tol = 2e-2 if mode == "bf16" else TOL
        check(torch.where(torch.isfinite(logits), logits, torch.zeros_like(logits)), np.where(fin, l_ref, 0), "logits V=%d A=%d %s" % (V, A, mode), tol)
        check(p, p_ref, "p V=%d A=%d %s" % (V, A, mode), tol)
    finally:
        cti_amd.set_precision(old)


def test_empty_batch_returns_empty_tensors():
    att = cti_amd.TriAttention(16, 16, 16, 16, 1, 4, 2, 1).to(DEV).eval()
    net = cti_amd.TCNet(16, 16, 16, 16, 1, 4, 1, k=2).to(DEV).eval()
    bi = cti_amd.BiAttention(16, 16, 16, 2).to(DEV).eval()
    v, q, a = torch.zeros(0, 5, 16, device=DEV), torch.zeros(0, 3, 16, device=DEV), torch.zeros(0, 2, 16, device=DEV)
    with torch.no_grad():
        p, logits = att(v, q, a)
        assert tuple(p.shape) == (0, 5, 3, 2, 2) and tuple(logits.shape) == (0, 5, 3, 2, 2)
        assert tuple(net.forward_with_weights(v, q, a, p[..., 0]).shape) == (0, 32)
        pb, lb = bi.forward_all(v, q)
        assert tuple(pb.shape) == (0, 2, 5, 3)


def test_non_contiguous_inputs():
    torch.manual_seed(3)
    att = cti_amd.TriAttention(24, 16, 16, 32, 1, 4, 2, 1).to(DEV).eval()
    rs = np.random.RandomState(9)
    vbig = np.abs(rs.standard_normal((3, 10, 48))).astype(np.float32)
    qbig = rs.standard_normal((3, 4, 2, 16)).astype(np.float32)
    a = rs.standard_normal((3, 2, 16)).astype(np.float32)
    v_t, q_t = T(vbig)[:, ::2, 12:36], T(qbig)[:, :, 1, :]                 # strided rows, offset columns
    with torch.no_grad():
        p, _ = att(v_t, q_t, T(a))
    p_ref, _ = O.tri_attention(vbig[:, ::2, 12:36], qbig[:, :, 1, :], a, sd(att), dtype=np.float64)
    check(p, p_ref, "p (non-contiguous inputs)")


# ---- degenerate shapes of the training-side kernels added with the fused rank nets / streaming mode-3 backward -------------------------
@pytest.mark.parametrize("R,hr,h,rows", [(1, 1, 4, 1), (2, 16, 512, 1), (3, 5, 8, 17), (1, 16, 64, 129)])
def test_rank_net_kernels_at_degenerate_sizes(R, hr, h, rows):
    """cti_ranknets_drop_fwd / _dw / _dx against float64 with the mask applied by hand: one row, one rank, hr = 1, a row count one past a tile."""
    ops = cti_amd.pkg.ops
    g = torch.Generator().manual_seed(R * 1000 + rows)
    x = torch.randn(rows, h, generator=g); W = torch.randn(R * hr, h, generator=g) / 4
    scale = torch.rand(R, generator=g) + 0.5; bias = torch.randn(R * hr, generator=g)
    dzs = torch.randn(rows, R * hr, generator=g)
    p = 0.4
    mask = ops.dropout_mask((R, rows, h), p, torch.device(DEV))
    mk = mask.cpu().double() / (1 - p)                                                        # (R, rows, h)
    Xd = x.double()[None] * mk
    y = ops.ranknets_drop_fwd(x.to(DEV), mask, W.to(DEV), scale.to(DEV), bias.to(DEV), R, p, True)
    assert y is not None
    ref = torch.cat([torch.relu(scale[r].double() * (Xd[r] @ W[r * hr:(r + 1) * hr].double().t()) + bias[r * hr:(r + 1) * hr].double()) for r in range(R)], 1)
    check(y, ref.numpy(), "rank nets fwd", 2e-5)
    G = ops.ranknets_drop_dw(dzs.to(DEV), x.to(DEV), mask, R, p)
    refG = torch.cat([dzs[:, r * hr:(r + 1) * hr].double().t() @ Xd[r] for r in range(R)], 0)
    check(G, refG.numpy(), "rank nets dW", 2e-5)
    dx = ops.ranknets_drop_dx(dzs.to(DEV), W.to(DEV), mask, R, p)
    refx = sum(mk[r] * (dzs[:, r * hr:(r + 1) * hr].double() @ W[r * hr:(r + 1) * hr].double()) for r in range(R))
    check(dx, refx.numpy(), "rank nets dx", 2e-5)


@pytest.mark.parametrize("B,V,Q,A,G,K", [(1, 1, 1, 1, 1, 4), (1, 1, 1, 8, 1, 32), (2, 64, 16, 3, 2, 64)])
def test_core_backward_at_degenerate_sizes(B, V, Q, A, G, K):
    g = torch.Generator().manual_seed(B + V + K)
    dout = torch.randn(B, V, Q, A, G, generator=g); M = torch.randn(B, V, Q, G, K, generator=g); Ar = torch.randn(B, A, K, generator=g)
    dM, dAr = cti_amd.pkg.ops.paralind_core_bwd(dout.to(DEV), M.to(DEV), Ar.to(DEV))
    check(dM, torch.einsum("bvqag,bak->bvqgk", dout.double(), Ar.double()).numpy(), "dM", 1e-5)
    check(dAr, torch.einsum("bvqag,bvqgk->bak", dout.double(), M.double()).numpy(), "dAr", 1e-5)


@pytest.mark.parametrize("B,G,V,Q,D", [(1, 1, 1, 1, 32), (2, 8, 64, 16, 64), (3, 2, 17, 5, 96), (2, 3, 36, 14, 1056)])
def test_bilinear_logits_and_bi_pool_backward_at_tile_limits(B, G, V, Q, D):
    """The LDS-staged bilinear logits (V <= 64, G*Q <= 128) and the register-resident bi-pool backward (Q <= 16) at their bounds and at 1."""
    ops = cti_amd.pkg.ops
    g = torch.Generator().manual_seed(B * 7 + D)
    vt = torch.randn(B, V, D, generator=g); qt = torch.randn(B, Q, D, generator=g); h = torch.randn(G, D, generator=g) / 8
    hb = torch.randn(G, generator=g); hs = torch.tensor([0.7])
    lg = ops.bi_logits(vt.to(DEV), qt.to(DEV), h.to(DEV), hs.to(DEV), hb.to(DEV))
    ref = 0.7 * torch.einsum("bvd,gd,bqd->bgvq", vt.double(), h.double(), qt.double()) + hb.double()[None, :, None, None]
    check(lg, ref.numpy(), "bi logits", 2e-5)
    w = torch.rand(B, V, Q, generator=g); dout = torch.randn(B, D, generator=g)
    dvt, dqt, dw = ops.bi_pool_bwd(dout.to(DEV), vt.to(DEV), qt.to(DEV), w.to(DEV), 1)
    check(dvt, (dout.double()[:, None, :] * torch.einsum("bvq,bqd->bvd", w.double(), qt.double())).numpy(), "bi pool dvt", 2e-5)
    check(dqt, (dout.double()[:, None, :] * torch.einsum("bvq,bvd->bqd", w.double(), vt.double())).numpy(), "bi pool dqt", 2e-5)
    check(dw, torch.einsum("bd,bvd,bqd->bvq", dout.double(), vt.double(), qt.double()).numpy(), "bi pool dw", 2e-5)


@pytest.mark.parametrize("hr,R,G", [(3, 5, 2), (12, 2, 3), (32, 2, 1)])
def test_mbuild_backward_for_any_core_size(hr, R, G):
    """h/rank outside {4, 8, 16} (the reference takes any --rank / --h_mm, src/FFOE/main.py:61-64): the generic VALU backward against float64
    autograd of the closed form M[b,v,q,g,r,k] = sum_ij T[r,i,j,k,g] Vr[b,v,r,i] Qr[b,q,r,j]."""
    ops = cti_amd.ops
    g = torch.Generator().manual_seed(hr * 100 + R)
    B, V, Q = 3, 5, 4
    Vr = torch.randn(B, V, R * hr, generator=g); Qr = torch.randn(B, Q, R * hr, generator=g)
    T_ = torch.randn(R, hr, hr, hr, G, generator=g); dM = torch.randn(B, V, Q, G, R * hr, generator=g)
    dVr, dQr, dT = ops.paralind_mbuild_bwd(dM.to(DEV), Vr.to(DEV), Qr.to(DEV), T_.to(DEV))
    v64, q64, t64 = (x.double().requires_grad_(True) for x in (Vr, Qr, T_))
    M = torch.einsum("rijkg,bvri,bqrj->bvqgrk", t64, v64.view(B, V, R, hr), q64.view(B, Q, R, hr)).reshape(B, V, Q, G, R * hr)
    (M * dM.double()).sum().backward()
    for got, ref, n_ in ((dVr, v64.grad, "dVr"), (dQr, q64.grad, "dQr"), (dT, t64.grad, "dT")):
        e = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
        assert e < 1e-5, (n_, e)
    # and the forward of the same shapes (generic kernel) for completeness
    Mf = ops.paralind_mbuild(Vr.to(DEV), Qr.to(DEV), T_.to(DEV))
    assert float((Mf.cpu().double() - M.detach()).abs().max() / M.abs().max()) < 1e-5


@pytest.mark.parametrize("B,V,Q,R", [(3, 36, 14, 32), (2, 36, 12, 4), (2, 1, 1, 2), (2, 48, 10, 2), (1, 17, 16, 3), (2, 37, 14, 2)])
@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_mbuild_backward_on_the_matrix_cores(B, V, Q, R, prec):
    """cti_paralind_mbuild_bwd_mfma (hr = 16, G = 2: the models' cores) against float64 autograd of the closed form, at the model shapes (V = 36, Q = 14 / 12),
    at one object / one token, at the tile limits (V = 48, Q = 16) and at a ragged V; and against the exact-fp32 VALU kernel it replaces."""
    ops = cti_amd.ops
    hr, G = 16, 2
    g = torch.Generator().manual_seed(V * 100 + Q + R)
    Vr = torch.randn(B, V, R * hr, generator=g); Qr = torch.randn(B, Q, R * hr, generator=g)
    T_ = torch.randn(R, hr, hr, hr, G, generator=g); dM = torch.randn(B, V, Q, G, R * hr, generator=g)
    old = ops.get_precision()
    try:
        ops.set_precision(prec)
        dVr, dQr, dT = ops.paralind_mbuild_bwd(dM.to(DEV), Vr.to(DEV), Qr.to(DEV), T_.to(DEV))
        ops.set_precision("fp32")
        eVr, eQr, eT = ops.paralind_mbuild_bwd(dM.to(DEV), Vr.to(DEV), Qr.to(DEV), T_.to(DEV))
    finally:
        ops.set_precision(old)
    v64, q64, t64 = (x.double().requires_grad_(True) for x in (Vr, Qr, T_))
    M = torch.einsum("rijkg,bvri,bqrj->bvqgrk", t64, v64.view(B, V, R, hr), q64.view(B, Q, R, hr)).reshape(B, V, Q, G, R * hr)
    (M * dM.double()).sum().backward()
    tol = 2e-5 if prec == "bf16x3" else 2e-2
    for got, exact, ref, n_ in ((dVr, eVr, v64.grad, "dVr"), (dQr, eQr, q64.grad, "dQr"), (dT, eT, t64.grad, "dT")):
        assert torch.isfinite(got).all(), n_
        e = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
        assert e < tol, (n_, e)
        e2 = float((exact.cpu().double() - ref).abs().max() / ref.abs().max())
        assert e2 < 1e-5, (n_, "exact", e2)


@pytest.mark.parametrize("R,rows", [(2, 1), (5, 130), (32, 77), (3, 64), (32, 300)])
@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_rank_net_forward_on_the_matrix_cores(R, rows, prec):
    """cti_ranknets_drop_fwd_mfma (h = 512, hr = 16: the models' widths) against float64 with the mask applied by hand and against the fp32-MFMA kernel it
    replaces: one row, row counts around the 64-row workgroup, one rank group and several (R = 32 splits over workgroups), with and without ReLU."""
    ops = cti_amd.pkg.ops
    hr, h, p = 16, 512, 0.3
    g = torch.Generator().manual_seed(R * 1000 + rows)
    x = torch.randn(rows, h, generator=g); W = torch.randn(R * hr, h, generator=g) / 4
    scale = torch.rand(R, generator=g) + 0.5; bias = torch.randn(R * hr, generator=g)
    mask = ops.dropout_mask((R, rows, h), p, torch.device(DEV))
    Xd = x.double()[None] * (mask.cpu().double() / (1 - p))
    old = ops.get_precision()
    try:
        for relu in (True, False):
            ops.set_precision(prec)
            y = ops.ranknets_drop_fwd(x.to(DEV), mask, W.to(DEV), scale.to(DEV), bias.to(DEV), R, p, relu)
            ops.set_precision("fp32")
            y32 = ops.ranknets_drop_fwd(x.to(DEV), mask, W.to(DEV), scale.to(DEV), bias.to(DEV), R, p, relu)
            ref = torch.cat([scale[r].double() * (Xd[r] @ W[r * hr:(r + 1) * hr].double().t()) + bias[r * hr:(r + 1) * hr].double() for r in range(R)], 1)
            if relu:
                ref = torch.relu(ref)
            den = float(ref.abs().max())
            assert float((y32.cpu().double() - ref).abs().max()) / den < 2e-6
            assert float((y.cpu().double() - ref).abs().max()) / den < (2e-5 if prec == "bf16x3" else 2e-2), (prec, relu)
    finally:
        ops.set_precision(old)


def test_matrix_core_backward_kernels_at_the_full_training_shapes():
    """Whole-launch coverage at the shapes of the data-parallel step (BASELINE configs[4]: 256 rows per GPU): the MFMA M-build backward (every (rank, chunk)
    workgroup: 32 x 8) and the MFMA rank-net forward (every 64-row workgroup and rank group of the visual branch, 9 216 rows) against the exact-fp32 kernels they
    replace -- all outputs, <= 2e-5 of the largest value."""
    ops = cti_amd.pkg.ops
    g = torch.Generator().manual_seed(77)
    B, V, Q, R, hr, G = 256, 36, 14, 32, 16, 2
    Vr = torch.randn(B, V, R * hr, generator=g).to(DEV); Qr = torch.randn(B, Q, R * hr, generator=g).to(DEV)
    T_ = torch.randn(R, hr, hr, hr, G, generator=g).to(DEV); dM = torch.randn(B, V, Q, G, R * hr, generator=g).to(DEV)
    rows, h, p = B * V, 512, 0.3
    x = torch.randn(rows, h, generator=g).to(DEV); W = (torch.randn(R * hr, h, generator=g) / 4).to(DEV)
    scale = (torch.rand(R, generator=g) + 0.5).to(DEV); bias = torch.randn(R * hr, generator=g).to(DEV)
    mask = ops.dropout_mask((R, rows, h), p, torch.device(DEV))
    old = ops.get_precision()
    try:
        ops.set_precision("bf16x3")
        got = ops.paralind_mbuild_bwd(dM, Vr, Qr, T_) + (ops.ranknets_drop_fwd(x, mask, W, scale, bias, R, p, True),)
        ops.set_precision("fp32")
        ref = ops.paralind_mbuild_bwd(dM, Vr, Qr, T_) + (ops.ranknets_drop_fwd(x, mask, W, scale, bias, R, p, True),)
    finally:
        ops.set_precision(old)
    for a, b_, n_ in zip(got, ref, ("dVr", "dQr", "dT", "rank nets y")):
        assert torch.isfinite(a).all(), n_
        e = float((a - b_).abs().max() / b_.abs().max())
        assert e < 2e-5, (n_, e)


@pytest.mark.parametrize("prec,tol", [("bf16", 8e-3), ("bf16x3", 2e-5)])
@pytest.mark.parametrize("M,n,N,K", [(9216, 3, 1024, 2048), (2304, 1, 3072, 2048), (777, 2, 1024, 512)])
def test_projection_gemm_every_row_at_model_shapes(prec, tol, M, n, N, K):
    """The hoisted projection GEMMs (cti_gemm_nt_pb: 256 x 256 tiles at these sizes) with EVERY output row checked against float64, per 64-row block of the
    tile: the plain-bf16 form of that geometry once launched with less LDS than its staged epilogue uses (the last waves' rows of each tile came out as bias
    only) and nothing looked at those rows -- the model-level checks compare the first samples."""
    ops = cti_amd.pkg.ops
    g = torch.Generator().manual_seed(M + n)
    a = torch.randn(M, K, generator=g).to(DEV); w = (torch.randn(n * N, K, generator=g) / 8).to(DEV); b = torch.randn(n * N, generator=g).to(DEV)
    old = ops.get_precision()
    try:
        ops.set_precision(prec)
        wp = ops.split_operand(w)
        y = ops.gemm_nt(a, w, nb1=n, rA1=0, rB1=N, M=M, N=N, bias=b, bias_bs=N, relu=True, B_planes=wp)
    finally:
        ops.set_precision(old)
    ref = torch.relu((a.double() @ w.double().t() + b.double()).view(M, n, N).permute(1, 0, 2))
    err = (y.double().view(n, M, N) - ref).abs().amax(-1) / ref.abs().max()                  # (n, M)
    worst = float(err.max())
    assert worst < tol, (prec, worst, int(err.argmax() % M))


@pytest.mark.parametrize("prec,tol", [("bf16", 1e-2), ("bf16x3", 2e-5), ("fp32", 5e-6)])
@pytest.mark.parametrize("rows,K,N,n_mats", [(9216, 2048, 512, 1), (9216, 512, 512, 32), (3072, 1024, 3072, 1)])
def test_linear_layer_forward_and_backward_every_entry_at_training_shapes(prec, tol, rows, K, N, n_mats):
    """WNLinearFn WITHOUT an activation (a ReLU kink would turn 1e-6 forward differences into single-entry gradient flips) at the row counts of the training step:
    y, dx (cti_gemm_nn), dV (cti_gemm_tn: split-K over the rows), dg and db -- every entry against float64 autograd.  These row counts select the 256-row tiles
    and the split-K planners that the small fixtures never reach."""
    AG = cti_amd.pkg.autograd
    g = torch.Generator().manual_seed(rows + N + n_mats)
    x = torch.randn(rows, K, generator=g).to(DEV); V = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    gg = (torch.rand(n_mats, generator=g) + 0.5).to(DEV); b = (torch.randn(N, generator=g) * 0.1).to(DEV); dy = torch.randn(rows, N, generator=g).to(DEV)
    x6, V6, g6, b6 = (t.double().requires_grad_(True) for t in (x, V, gg, b))
    sc = (g6 / V6.view(n_mats, -1).norm(dim=1)).repeat_interleave(N // n_mats)
    y6 = x6 @ V6.t() * sc + b6
    (y6 * dy.double()).sum().backward()
    old = cti_amd.get_precision()
    try:
        cti_amd.set_precision(prec)
        xs, Vs, gs, bs = (t.clone().requires_grad_(True) for t in (x, V, gg, b))
        y = AG.WNLinearFn.apply(xs, Vs, gs, bs, False, n_mats)
        (y * dy).sum().backward()
    finally:
        cti_amd.set_precision(old)
    for got, ref, n_ in ((y, y6.detach(), "y"), (xs.grad, x6.grad, "dx"), (Vs.grad, V6.grad, "dV"), (gs.grad, g6.grad, "dg"), (bs.grad, b6.grad, "db")):
        e = float((got.double() - ref).abs().max() / ref.abs().max())
        assert e < (10 * tol if n_ == "dg" else tol), (n_, e)               # dg = <G, V> / g: a scalar left over from a million cancelling terms


def test_pools_and_bilinear_logits_every_sample_at_full_batch():
    """The sum-pools and the bilinear logits at the FFOE / MC shapes with all 256 samples against float64 einsums on the device: these kernels are the same in every
    precision mode (fp32 VALU / their own MFMA forms), so the cross-precision model tests cannot see them, and the oracle checks of the bench line cover four samples."""
    ops = cti_amd.pkg.ops
    g = torch.Generator().manual_seed(9)
    B, V, D = 256, 36, 1024
    for (Q, A) in ((14, 3), (12, 6)):
        vt = torch.randn(B, V, D, generator=g).to(DEV); qt = torch.randn(B, Q, D, generator=g).to(DEV); at = torch.randn(B, A, D, generator=g).to(DEV)
        att = torch.softmax(torch.randn(B, V * Q * A, 2, generator=g), 1).view(B, V, Q, A, 2).to(DEV)
        for gl in (0, 1):
            w = att[..., gl]
            out = ops.tri_pool(vt, qt, at, w)
            ref = torch.einsum("bvd,bqd,bad,bvqa->bd", vt.double(), qt.double(), at.double(), w.double())
            e = ((out.double() - ref).abs().amax(1) / ref.abs().max())
            assert float(e.max()) < 2e-5, ("tri_pool", Q, A, gl, float(e.max()), int(e.argmax()))
    Q, D3, G = 14, 3072, 8
    vt = torch.randn(B, V, D3, generator=g).to(DEV); qt = torch.randn(B, Q, D3, generator=g).to(DEV)
    w = torch.softmax(torch.randn(B, G, V * Q, generator=g), 2).view(B, G, V, Q).to(DEV)
    out = ops.bi_pool(vt, qt, w[:, 3], 3)
    ref = torch.einsum("bvd,bqd,bvq->bd", vt.double(), qt.double(), w[:, 3].double()).view(B, D3 // 3, 3).sum(2)
    e = ((out.double() - ref).abs().amax(1) / ref.abs().max())
    assert float(e.max()) < 2e-5, ("bi_pool k=3", float(e.max()), int(e.argmax()))
    out = ops.bi_pool(vt[..., :1024].contiguous(), qt[..., :1024].contiguous(), w[:, 5], 1)
    ref = torch.einsum("bvd,bqd,bvq->bd", vt[..., :1024].double(), qt[..., :1024].double(), w[:, 5].double())
    e = ((out.double() - ref).abs().amax(1) / ref.abs().max())
    assert float(e.max()) < 2e-5, ("bi_pool k=1", float(e.max()), int(e.argmax()))
    h = (torch.randn(G, D3, generator=g) / 8).to(DEV); hb = torch.randn(G, generator=g).to(DEV); hs = torch.tensor([0.7], device=DEV)
    old = cti_amd.get_precision()
    try:
        for prec, tol in (("bf16x3", 2e-5), ("fp32", 5e-6)):
            cti_amd.set_precision(prec)
            lg = ops.bi_logits(vt, qt, h, hs, hb)
            ref = 0.7 * torch.einsum("bvd,gd,bqd->bgvq", vt.double(), h.double(), qt.double()) + hb.double()[None, :, None, None]
            e = ((lg.double() - ref).abs().flatten(1).amax(1) / ref.abs().max())
            assert float(e.max()) < tol, ("bi_logits", prec, float(e.max()), int(e.argmax()))
    finally:
        cti_amd.set_precision(old)
