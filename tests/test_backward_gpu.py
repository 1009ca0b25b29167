"""Gradient parity of the HIP backward path against the gradients the reference's autograd produced (tests/golden, `g/*`).
loss = sum(out * cot) with the cotangent stored in the fixture.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import cti_amd
import golden_util as gu
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


def T(x, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t.requires_grad_(True) if grad else t


def load_into(m, params):
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.to(DEV).eval()


def check(x, ref, tol=TOL, what=""):
    x = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x
    assert x.shape == tuple(ref.shape), (what, x.shape, ref.shape)
    e = O.norm_max_err(x, ref)
    assert e < tol, "%s: normalised max error %.3g >= %.3g" % (what, e, tol)


def check_param_grads(m, fx, prefix="p/", tol=TOL):
    """Tensor-valued gradients: normalised max error per parameter.  The 0-d weight-norm gains (`weight_g`, `h_mat_g`) are
    <G, V> / g, a cancelling sum whose own magnitude says nothing about its conditioning, so they are compared together as ONE
    vector (which is how the reference's Trainer sees them: one flat gradient buffer, src/FFOE/trainer.py:245-255)."""
    n = 0
    s_got, s_ref = [], []
    for name, p in m.named_parameters():
        key = prefix + name
        if key in fx.g:
            assert p.grad is not None, name
            if p.dim() == 0:
                s_got.append(float(p.grad)); s_ref.append(float(fx.g[key]))
            elif float(np.max(np.abs(fx.g[key]))) < 1e-6:
                # mathematically zero (h_bias under a softmax: the softmax is shift-invariant per row); the reference holds
                # 1e-8 rounding noise there, so only the absolute size can be checked
                assert float(p.grad.abs().max()) < 1e-5, name
            else:
                check(p.grad, fx.g[key], tol, "grad of " + name)
            n += 1
    assert n > 0
    if s_ref:
        check(np.array(s_got, np.float32), np.array(s_ref, np.float32), tol, "weight-norm gain gradients (as one vector)")


@pytest.mark.parametrize("name", ["g0_fcnet_2layer", "g0_fcnet_noact", "g0_fcnet_drop"])
def test_fcnet_backward(name):
    fx = gu.load(name)
    c = fx.cfg
    m = load_into(cti_amd.FCNet(c["dims"], act=c["act"], dropout=c["dropout"]), fx.p)
    x = T(fx.i["x"], grad=True)
    y = m(x)
    check(y, fx.o["y"], what=name + " forward under autograd")
    (y * T(fx.i["cot"])).sum().backward()
    check(x.grad, fx.g["x"], what=name + " dx")
    check_param_grads(m, fx)


def test_gemm_tn_split_k_vs_numpy():
    rs = np.random.RandomState(0)
    for M, N, K in ((5, 3, 4), (300, 70, 33), (5000, 130, 64), (9216, 96, 40)):
        a = rs.standard_normal((M, N)).astype(np.float32)
        b = rs.standard_normal((M, K)).astype(np.float32)
        ref = a.astype(np.float64).T @ b.astype(np.float64)
        out = cti_amd.ops.gemm_tn(T(a), T(b))
        check(out, ref, tol=2e-5, what="gemm_tn %dx%dx%d" % (M, N, K))


def test_dropout_mask_statistics_and_backward():
    torch.manual_seed(3)
    x = torch.randn(64, 1000, device=DEV, requires_grad=True)
    y = cti_amd.pkg.autograd.dropout(x, 0.3, True)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01
    assert torch.allclose(y[y != 0], (x / 0.7)[y != 0])
    y.sum().backward()
    assert torch.equal(x.grad != 0, y != 0) and torch.allclose(x.grad[y != 0], torch.full_like(x.grad[y != 0], 1 / 0.7))
    y2 = cti_amd.pkg.autograd.dropout(x, 0.3, True)                  # a fresh mask every call
    assert not torch.equal(y2 != 0, y != 0)
    assert cti_amd.pkg.autograd.dropout(x, 0.3, False) is x          # eval mode: identity


def _tri(fx):
    c = fx.cfg
    return load_into(cti_amd.TriAttention(c["v_dim"], c["q_dim"], c["a_dim"], c["h_dim"], 1, c["rank"], c["glimpse"], c["k"]), fx.p)


@pytest.mark.parametrize("name", ["g3_tcnet_small", "g3_tcnet_g3_odd"])
def test_tcnet_forward_backward(name):
    fx = gu.load(name)
    m = _tri(fx)
    v, q, a = T(fx.i["v"], True), T(fx.i["q"], True), T(fx.i["a"], True)
    raw = m.TriAtt(v, q, a)
    check(raw, fx.o["raw"], what=name + " forward under autograd")
    (raw * T(fx.i["cot_raw"])).sum().backward()
    for n_, t_ in (("v", v), ("q", q), ("a", a)):
        check(t_.grad, fx.g[n_], what="%s d%s" % (name, n_))
    check_param_grads(m, fx)


def test_triattention_backward_through_softmax():
    fx = gu.load("g8_triattention_grad")
    m = _tri(fx)
    v, q, a = T(fx.i["v"], True), T(fx.i["q"], True), T(fx.i["a"], True)
    p, logits = m(v, q, a)
    check(p, fx.o["p"], what="p under autograd")
    assert np.array_equal(np.isneginf(logits.detach().cpu().numpy()), np.isneginf(fx.o["logits"]))
    (p * T(fx.i["cot_p"])).sum().backward()
    for n_, t_ in (("v", v), ("q", q), ("a", a)):
        check(t_.grad, fx.g[n_], what="d%s" % n_)
    check_param_grads(m, fx)


@pytest.mark.parametrize("name", ["g5_tcnet_fww_k2", "g5_tcnet_fww_k1"])
def test_tcnet_forward_with_weights_backward(name):
    fx = gu.load(name)
    c = fx.cfg
    m = load_into(cti_amd.TCNet(c["v_dim"], c["q_dim"], c["a_dim"], c["h_dim"], 1, c["rank"], c["glimpse"], dropout=[.2, .5], k=c["k"]), fx.p)
    v, q, a, att = T(fx.i["v"], True), T(fx.i["q"], True), T(fx.i["a"], True), T(fx.i["att"], True)
    out = m.forward_with_weights(v, q, a, att[:, :, :, :, 1])
    check(out, fx.o["out_g1"], what=name)
    (out * T(fx.i["cot"])).sum().backward()
    for n_, t_ in (("v", v), ("q", q), ("a", a), ("att", att)):
        check(t_.grad, fx.g[n_], what="%s d%s" % (name, n_))
    check_param_grads(m, fx)


@pytest.mark.parametrize("name", ["g6_bcnet_hnone_k1", "g6_bcnet_h2_k3", "g6_bcnet_h40_k1"])
def test_bcnet_backward(name):
    fx = gu.load(name)
    c = fx.cfg
    m = load_into(cti_amd.BCNet(c["v_dim"], c["q_dim"], c["h_dim"], c["h_out"], k=c["k"]), fx.p)
    v, q = T(fx.i["v"], True), T(fx.i["q"], True)
    out = m(v, q)
    check(out, fx.o["fwd"], what=name + " forward")
    (out * T(fx.i["cot_fwd"])).sum().backward()
    check(v.grad, fx.g["fwd/v"], what=name + " fwd dv")
    check(q.grad, fx.g["fwd/q"], what=name + " fwd dq")
    check_param_grads(m, fx, prefix="fwd/p/")
    m.zero_grad()
    v, q, w = T(fx.i["v"], True), T(fx.i["q"], True), T(fx.i["w"], True)
    out = m.forward_with_weights(v, q, w[:, 1])
    check(out, fx.o["fww"], what=name + " fww")
    (out * T(fx.i["cot_fww"])).sum().backward()
    check(v.grad, fx.g["fww/v"], what=name + " fww dv")
    check(q.grad, fx.g["fww/q"], what=name + " fww dq")
    check(w.grad, fx.g["fww/w"], what=name + " fww dw")
    check_param_grads(m, fx, prefix="fww/p/")


@pytest.mark.parametrize("name", ["g7_biattention_g2", "g7_biattention_g8", "g7_biattention_nomask"])
def test_biattention_backward(name):
    fx = gu.load(name)
    c = fx.cfg
    m = load_into(cti_amd.BiAttention(c["x_dim"], c["y_dim"], c["z_dim"], c["glimpse"]), fx.p)
    v, q = T(fx.i["v"], True), T(fx.i["q"], True)
    p, logits = m.forward_all(v, q, c["v_mask"])
    check(p, fx.o["p"], what=name + " p")
    (p * T(fx.i["cot_p"])).sum().backward()
    check(v.grad, fx.g["v"], what=name + " dv")
    check(q.grad, fx.g["q"], what=name + " dq")
    check_param_grads(m, fx)


def test_train_mode_runs_with_dropout_and_gives_gradients_to_every_parameter():
    """The reference's Trainer raises if any trainable parameter has no gradient (src/FFOE/trainer.py:239-241)."""
    fx = gu.load("g3_tcnet_small")
    m = _tri(fx).train()
    v, q, a = T(fx.i["v"]), T(fx.i["q"], True), T(fx.i["a"], True)
    torch.manual_seed(0)
    p, logits = m(v, q, a)
    assert p.shape == fx.o["p"].shape
    ok = ~torch.isnan(p)
    p[ok].sum().backward() if False else (p.nan_to_num() * torch.randn_like(p)).sum().backward()
    for n_, prm in m.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), n_
    assert q.grad is not None and a.grad is not None


@pytest.mark.parametrize("R,hr,h,lead,relu,fused", [
    (4, 16, 64, (5, 14), True, True),          # the original case
    (32, 16, 512, (3, 36), True, True),        # the real widths (h_mm 512, rank 32)
    (3, 6, 24, (7, 3), False, True),           # hr not a multiple of 4 (scalar dzs path), h < one 64-column block, odd rows
    (2, 4, 132, (1, 67), True, True),          # h not a multiple of 16 / 64: ragged k tails; rows not a multiple of 16
    (5, 12, 260, (2, 9), True, True),          # NS = 32 template with a tail
    (3, 8, 80, (2, 21), True, True),           # h % 16 == 0 (16-B mask loads, permuted k order) with a partial 64-column group
    (2, 16, 272, (1, 40), False, True),        # the same in the NS = 32 template
    (2, 32, 64, (3, 5), True, False),          # hr > 16: the general route (masked copies + batched GEMMs)
    (2, 8, 576, (2, 4), True, False),          # h > 512: the general route
])
def test_batched_rank_nets_under_dropout_match_a_manual_per_rank_evaluation(R, hr, h, lead, relu, fused):
    """RankNetsDropFn (train mode; fused mask-on-fragment kernels or the general route) against the same masks applied rank by rank
    with the plain WNLinear Function."""
    torch.manual_seed(5)
    AG = cti_amd.pkg.autograd
    x = torch.randn(*lead, h, device=DEV, requires_grad=True)
    wv = (torch.randn(R * hr, h, device=DEV) / 8).requires_grad_(True)
    g = (torch.rand(R, device=DEV) + 0.5).requires_grad_(True)
    b = (torch.randn(R * hr, device=DEV) / 10).requires_grad_(True)
    y = AG.RankNetsDropFn.apply(x, wv, g, b, relu, R, 0.5)
    # the masks are among the saved tensors (before backward frees them)
    mask = y.grad_fn.saved_tensors[1].clone().view(R, *lead, h)
    assert (y.grad_fn.saved_tensors[0].numel() == x.numel()) == fused        # fused: the plain input is saved, not R masked copies
    outs = []
    for r in range(R):
        xr = x * (mask[r].float() / 0.5)
        outs.append(AG.WNLinearFn.apply(xr, wv[r * hr:(r + 1) * hr], g[r], b[r * hr:(r + 1) * hr], relu, 1))
    y2 = torch.cat(outs, -1)
    check(y, y2.detach().cpu().numpy(), tol=2e-5, what="batched rank nets forward")
    # the two routes round differently (exact fp32 vs split bf16): an output within ~1e-6 of zero can sit on opposite sides of the ReLU, and
    # the gradients then legitimately differ there -- no cotangent on such elements
    cot = torch.randn_like(y) * ((y > 0) == (y2 > 0)).float()
    (y * cot).sum().backward()
    got = [t.grad.clone() for t in (x, wv, g, b)]
    for t in (x, wv, g, b):
        t.grad = None
    (y2 * cot).sum().backward()
    for n_, a_, t in zip(("dx", "dwv", "dg", "db"), got, (x, wv, g, b)):
        check(a_, t.grad.cpu().numpy(), tol=1e-4, what="batched rank nets " + n_)
    keep = mask.float().mean().item()
    assert abs(keep - 0.5) < 0.05


def _grads(model_fn, inputs, mode):
    cti_amd.set_precision(mode)
    xs = [t.detach().clone().requires_grad_(True) for t in inputs]
    out = model_fn(*xs)
    cot = torch.linspace(-1.0, 1.0, out.numel(), device=out.device).view_as(out)
    return out.detach(), xs, (out * cot).sum()


def test_model_width_backward_agrees_between_arithmetic_modes():
    """At the reference's real widths (D = 3072 / 1024, V = 36, Q = 14, G = 8) the MFMA forms of the backward kernels (bilinear logits, pool
    attention gradients, row-axis weight gradients) against the exact-fp32 mode of the same library: outputs and input gradients within
    1e-4 (max norm), parameter gradients within 1e-3 in the L2 sense."""
    if cti_amd.get_precision() != "bf16x3":
        pytest.skip("compares the two modes itself")
    torch.manual_seed(21)
    bi = cti_amd.BiAttention(2048, 1024, 1024, 8, dropout=[0.0, 0.0]).to(DEV)
    bnet = cti_amd.BCNet(2048, 1024, 1024, None, k=1, dropout=[0.0, 0.0]).to(DEV)
    tnet = cti_amd.TCNet(2048, 1024, 1024, 512, 1, 32, 1, k=2, dropout=[0.0, 0.0]).to(DEV)
    g = torch.Generator().manual_seed(5)
    v = torch.randn(4, 36, 2048, generator=g).abs().to(DEV); v[0, 30:] = 0
    q = torch.tanh(torch.randn(4, 14, 1024, generator=g)).to(DEV)
    a = torch.tanh(torch.randn(4, 3, 1024, generator=g)).to(DEV)
    w3 = torch.softmax(torch.randn(4, 36 * 14 * 3, generator=g), 1).view(4, 36, 14, 3).to(DEV)

    def f(v_, q_, a_, w_):
        p, _ = bi.forward_all(v_, q_)
        return torch.cat([p.flatten(1), bnet.forward_with_weights(v_, q_, p[:, 3]), tnet.forward_with_weights(v_, q_, a_, w_)], 1)

    res = {}
    for mode in ("fp32", "bf16x3"):
        for m in (bi, bnet, tnet):
            m.zero_grad(set_to_none=True)
        out, xs, loss = _grads(f, (v, q, a, w3), mode)
        loss.backward()
        res[mode] = (out, [x.grad.clone() for x in xs], {n: p_.grad.clone() for mod, tag in ((bi, "bi."), (bnet, "b."), (tnet, "t.")) for n, p_ in
                                                         ((tag + k, pp) for k, pp in mod.named_parameters()) if p_.grad is not None})
    cti_amd.set_precision("bf16x3")
    o32, g32, p32 = res["fp32"]
    o3, g3, p3 = res["bf16x3"]
    check(o3, o32.cpu().numpy(), what="outputs")
    for i, name in enumerate(("v", "q", "a", "w")):
        check(g3[i], g32[i].cpu().numpy(), what="d " + name)
    gains_a, gains_b = [], []
    for k in p32:
        if p32[k].dim() == 0:
            gains_a.append(float(p3[k])); gains_b.append(float(p32[k]))
        elif k.endswith("h_bias"):
            continue                     # mathematically zero (the softmax is shift-invariant per (b, g) row): both modes hold rounding noise
        elif float(p32[k].abs().max()) > 1e-6:
            # parameter gradients behind a ReLU: a pre-activation within rounding distance of zero flips its mask between the two modes
            # and changes single terms of the sums, so these are compared in the L2 sense (the smooth quantities above: max norm)
            rel = float((p3[k] - p32[k]).norm() / p32[k].norm())
            assert rel < 1e-3, "d %s: relative L2 difference %.3g" % (k, rel)
    ga, gb = np.array(gains_a), np.array(gains_b)
    assert np.linalg.norm(ga - gb) / np.linalg.norm(gb) < 1e-3, "weight-norm gains"


@pytest.mark.parametrize("B,V,Q,A,G,K", [(3, 5, 4, 3, 2, 64), (2, 36, 14, 6, 2, 512), (4, 7, 3, 8, 1, 132), (2, 3, 2, 3, 3, 20),
                                          (2, 4, 3, 9, 2, 32), (2, 4, 3, 3, 2, 30)])
def test_core_backward_streaming_passes_match_the_einsum(B, V, Q, A, G, K):
    """cti_paralind_core_bwd (and, for A > 8 or K % 4 != 0, the transposed-GEMM route behind the same op) against float64 einsums."""
    g = torch.Generator().manual_seed(B * 100 + K)
    dout = torch.randn(B, V, Q, A, G, generator=g)
    M = torch.randn(B, V, Q, G, K, generator=g)
    Ar = torch.randn(B, A, K, generator=g)
    dM, dAr = cti_amd.pkg.ops.paralind_core_bwd(dout.to(DEV), M.to(DEV), Ar.to(DEV))
    check(dM, torch.einsum("bvqag,bak->bvqgk", dout.double(), Ar.double()).numpy(), tol=1e-5, what="dM")
    check(dAr, torch.einsum("bvqag,bvqgk->bak", dout.double(), M.double()).numpy(), tol=1e-5, what="dAr")


@pytest.mark.parametrize("B,V,Q,A,R,hr,G", [(3, 36, 14, 3, 32, 16, 2), (2, 7, 5, 6, 8, 4, 1), (2, 10, 3, 3, 2, 16, 2), (2, 5, 4, 4, 4, 8, 3)])
def test_training_core_on_planes_matches_the_fp32_M_route(B, V, Q, A, R, hr, G):
    """MBuildCoreFn (M only as bf16 hi/lo planes; MFMA and VALU plane-writing M builds) against MBuildFn + CoreFn: outputs and all four gradients."""
    AG = cti_amd.pkg.autograd
    g = torch.Generator().manual_seed(R * 10 + hr)
    mk = lambda *s: (torch.randn(*s, generator=g) / 2).to(DEV).requires_grad_(True)
    Vr, Qr, Ar, Teff = mk(B, V, R * hr), mk(B, Q, R * hr), mk(B, A, R * hr), mk(R, hr, hr, hr, G)
    if not AG.MBuildCoreFn.supported(Vr, Ar, Teff):
        pytest.skip("the exact-fp32 mode keeps M in fp32")
    cot = torch.randn(B, V, Q, A, G, generator=g).to(DEV)
    res = []
    for fused in (True, False):
        out = AG.MBuildCoreFn.apply(Vr, Qr, Teff, Ar) if fused else AG.CoreFn.apply(AG.MBuildFn.apply(Vr, Qr, Teff), Ar)
        (out * cot).sum().backward()
        res.append([out.detach().cpu().numpy()] + [t.grad.cpu().numpy() for t in (Vr, Qr, Teff, Ar)])
        for t in (Vr, Qr, Teff, Ar):
            t.grad = None
    for n_, a_, b_ in zip(("out", "dVr", "dQr", "dTeff", "dAr"), res[0], res[1]):
        check(torch.from_numpy(a_), b_, tol=2e-5, what="core on planes " + n_)


@pytest.mark.parametrize("n", [16, 4096 + 16 * 3, 100003, 37])
def test_mask_only_dropout_draws_the_same_stream(n):
    """cti_dropout(y = NULL) (16 mask bytes per thread, or the 4-byte path for a tail / tiny input) draws exactly the mask the y-writing kernel
    draws from the same seed."""
    ops = cti_amd.pkg.ops
    x = torch.randn(n, device=DEV)
    ops.dropout_mask((16,), 0.5, x.device)              # settle the stream on the current torch seed (a new seed restarts the call counter)
    ops._dropout_calls[0] = 1234
    _, m1 = ops.dropout(x, 0.3)
    ops._dropout_calls[0] = 1234
    m2 = ops.dropout_mask((n,), 0.3, x.device)
    assert torch.equal(m1, m2)
    assert abs(float(m2.float().mean()) - 0.7) < (0.2 if n < 100 else 0.03)


def test_tcnet_with_another_activation_forward_and_backward():
    """The reference builds every FCNet activation by name (src/fc.py:24: getattr(nn, act)); TCNet(act='Tanh') takes the per-rank-net route
    (no packed single-GEMM form) and must still match the reference's output and gradients (fixture g3_tcnet_act_tanh)."""
    fx = gu.load("g3_tcnet_act_tanh")
    c = fx.cfg
    m = load_into(cti_amd.TCNet(c["v_dim"], c["q_dim"], c["a_dim"], c["h_dim"], 1, c["rank"], c["glimpse"], act=c["act"]), fx.p)
    v, q, a = T(fx.i["v"], True), T(fx.i["q"], True), T(fx.i["a"], True)
    with torch.no_grad():
        check(m(v.detach(), q.detach(), a.detach()), fx.o["raw"], what="TCNet(act=Tanh) eval forward")
    raw = m(v, q, a)
    check(raw, fx.o["raw"], what="TCNet(act=Tanh) forward under autograd")
    (raw * T(fx.i["cot_raw"])).sum().backward()
    for n_, t_ in (("v", v), ("q", q), ("a", a)):
        check(t_.grad, fx.g[n_], what="TCNet(act=Tanh) d%s" % n_)
    check_param_grads(m, fx)
