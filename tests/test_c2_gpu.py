"""Parity at the shape the headline metric is quoted on -- BASELINE.json configs[1]: TCNet.forward (reference src/tc.py:41-52) +
TriAttention (src/attention.py:49-59) at V=36x2048, Q=14x1024, A=3129x300, rank 32, h_mm 512, glimpse 2.

 (a) B=3 against the reference's own output (fixture g3_tcnet_forward_c2: 65 536 sampled positions of raw / p, -inf pattern, per-(b,g)
     argmax, log-sum-exp over the 1.58 M positions of a sample) and against the float64 oracle on every element;
 (b) B=256 once -- the launch bench.py times: the 256x256 `EPI_INTERLEAVE2` GEMM instantiation, 64-bit offsets of the 3.2 GB output, the
     49-chunk Tri softmax -- checking samples 0 / 127 / 255 against the float64 oracle;
 (c) every tile geometry of the plane GEMM and the multi-chunk softmax (forward and backward) forced at small shapes through
     cti_set_tuning, against the reference fixtures.

Tolerance: 1e-4 normalised max error (north_star); masks, -inf patterns and argmax bit-exact."""
import numpy as np
import pytest
import torch

import cti_amd
import golden_util as gu
from oracle import cti_oracle as O
from test_oracle_golden import c2_checks
from test_parity_gpu import T, load_into, check, _tri, TOL
from test_backward_gpu import check_param_grads

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tri_from(cfg, params):
    return _tri(type("F", (), {"cfg": cfg, "p": params})())


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "f16f6"])
def test_c2_widths_small_batch(prec):
    fx, params, v, q, a, idx = gu.c2_case()
    old = cti_amd.get_precision()
    cti_amd.set_precision(prec)
    try:
        m = _tri_from(fx.cfg, params)
        with torch.no_grad():
            raw = m.TriAtt(T(v), T(q), T(a)).cpu().numpy()
            p, logits = m(T(v), T(q), T(a))
        p, logits = p.cpu().numpy(), logits.cpu().numpy()
    finally:
        cti_amd.set_precision(old)
    e_raw, e_p = c2_checks(raw, p, logits, fx, idx, TOL, 1e-2, "HIP C2 [%s] vs reference" % prec)
    raw64 = O.tcnet_forward(v, q, a, params, "TriAtt.", dtype=np.float64)
    e64 = check(raw, raw64, what="C2 raw vs float64 oracle (all 9.5 M elements)")
    assert np.array_equal(np.broadcast_to(O.zero_row_mask(v).astype(bool)[:, :, None, None, None], raw.shape), np.isneginf(logits))
    print("C2 widths B=3 [%s]: raw vs reference sample %.3g, vs float64 truth %.3g" % (prec, e_raw, e64))


def _c2_batch(B, seed):
    rs = np.random.RandomState(seed)
    v = gu.rs_fill(rs, (B, 36, 2048), "abs")
    nv = rs.randint(10, 37, size=B)
    for b in range(B):
        v[b, nv[b]:] = 0
    q = gu.rs_fill(rs, (B, 14, 1024), "scale:1.0")
    a = gu.rs_fill(rs, (B, 3129, 300), "scale:1.0")
    return v, q, a


@pytest.mark.parametrize("prec", ["bf16x3", "f16f6"])
def test_c2_full_batch_sampled_against_oracle(prec):
    """The benchmarked launch itself (B=256): samples 0, 127, 255 against the float64 oracle."""
    old = cti_amd.get_precision()
    cti_amd.set_precision(prec)
    try:
        _c2_full_batch(prec)
    finally:
        cti_amd.set_precision(old)


def _c2_full_batch(prec):
    fx, params, *_ = gu.c2_case()
    B = 256
    v, q, a = _c2_batch(B, 777)
    v[127] = gu.rs_fill(np.random.RandomState(5), (36, 2048), "abs")      # no padding at all on sample 127; sample 255 gets the maximum padding
    v[255, 10:] = 0
    m = _tri_from(fx.cfg, params)
    pick = [0, 127, 255]
    with torch.no_grad():
        vd, qd, ad = T(v), T(q), T(a)
        raw = m.TriAtt(vd, qd, ad)
        assert raw.shape == (B, 36, 14, 3129, 2) and raw.is_contiguous()
        raw_s = raw[pick].cpu().numpy()
        del raw
        p, logits = m(vd, qd, ad)
        p_s, l_s = p[pick].cpu().numpy(), logits[pick].cpu().numpy()
        psum = p.view(B, -1, 2).sum(1).cpu().numpy()
        # whole-launch statistics: per (b, g) the maximum and the log-sum-exp over the sample's 1.58 M positions (masked rows are -inf) --
        # every tile of the persistent walk feeds one of these 512 pairs
        lg = logits.view(B, -1, 2)
        gmax = lg.max(1).values.cpu().numpy()
        glse = torch.logsumexp(lg.double(), 1).cpu().numpy()
        del p, logits, lg
    torch.cuda.empty_cache()
    # ... against the float32 oracle on ALL 256 samples (the reference's own arithmetic class; ~20 s on the GPU box's host cores)
    omax, olse, oabs = np.empty((B, 2)), np.empty((B, 2)), 0.0
    for b0 in range(0, B, 8):
        sl = slice(b0, b0 + 8)
        r32 = O.tcnet_forward(v[sl], q[sl], a[sl], params, "TriAtt.")
        oabs = max(oabs, float(np.max(np.abs(r32))))
        r32 = np.where(O.zero_row_mask(v[sl]).astype(bool)[:, :, None, None, None], -np.inf, r32).reshape(r32.shape[0], -1, 2).astype(np.float64)
        omax[sl] = r32.max(1)
        mx = omax[sl][:, None, :]
        olse[sl] = np.log(np.exp(r32 - mx).sum(1)) + omax[sl]
        del r32
    e_max, e_lse = np.max(np.abs(gmax - omax)) / oabs, np.max(np.abs(glse - olse)) / oabs
    assert e_max < TOL and e_lse < TOL, "B=256 [%s]: per-(b,g) max %.3g / log-sum-exp %.3g vs the float32 oracle over all samples" % (prec, e_max, e_lse)
    print("C2 B=256 [%s]: all 256 samples, per-(b,g) max err %.3g, log-sum-exp err %.3g (normalised by max |raw| = %.3g)" % (prec, e_max, e_lse, oabs))
    raw64 = O.tcnet_forward(v[pick], q[pick], a[pick], params, "TriAtt.", dtype=np.float64)
    e = check(raw_s, raw64, what="B=256 raw (samples 0/127/255) vs float64 oracle")
    p64, l64 = O.tri_attention(v[pick], q[pick], a[pick], params, dtype=np.float64)
    assert np.array_equal(np.isneginf(l_s), np.isneginf(l64))
    assert np.max(np.abs(psum - 1.0)) < 1e-4                       # every sample's 49-chunk softmax normalises
    for i in range(3):
        for g in range(2):
            ref = p64[i, ..., g].reshape(-1)
            top = np.argsort(ref)[-2:]
            if ref[top[1]] - ref[top[0]] > 1e-3:                    # a clear winner in float64: the fp32-grade result must agree
                assert int(np.argmax(p_s[i, ..., g])) == int(top[1])
    ok = np.isfinite(l64)
    assert np.max(np.abs(l_s[ok] - l64[ok])) < TOL * np.max(np.abs(l64[ok]))
    print("C2 B=256 [%s]: raw err vs float64 truth on samples 0/127/255 = %.3g" % (prec, e))


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_forced_tile_geometry_and_multichunk_softmax_forward(cfg, prec):
    """g3 fixtures with the GEMM tile geometry forced and the Tri softmax cut into 16-position chunks (4 and 7 chunks)."""
    old = cti_amd.get_precision()
    cti_amd.set_precision(prec)
    tol = TOL if prec == "bf16x3" else 3e-2
    try:
        for name in ("g3_tcnet_small", "g3_tcnet_g3_odd"):
            fx = gu.load(name)
            m = _tri(fx)
            v, q, a = T(fx.i["v"]), T(fx.i["q"]), T(fx.i["a"])
            with cti_amd.ops.tuning(gemm_cfg=cfg, tri_chunk=16), torch.no_grad():
                raw = m.TriAtt(v, q, a)
                p, logits = m(v, q, a)
            check(raw, fx.o["raw"], tol=tol, what="%s raw cfg=%d" % (name, cfg))
            assert np.array_equal(np.isneginf(logits.cpu().numpy()), np.isneginf(fx.o["logits"]))
            if prec == "bf16x3":
                check(p, fx.o["p"], what="%s p cfg=%d" % (name, cfg))
    finally:
        cti_amd.set_precision(old)
    assert cti_amd.pkg._lib.lib().cti_get_tuning(1) == -1 and cti_amd.pkg._lib.lib().cti_get_tuning(2) == 0


def test_c1_with_every_tile_geometry():
    """BASELINE configs[0] at full size with each geometry forced: the 256x256 tile on 1008 x 4 per-sample outputs, 128x128 on the rest."""
    fx, params, v, q, a = gu.c1_case()
    m = _tri_from(fx.cfg, params)
    for cfg in (0, 1, 2):
        with cti_amd.ops.tuning(gemm_cfg=cfg, tri_chunk=512), torch.no_grad():
            raw = m.TriAtt(T(v), T(q), T(a))
            p, logits = m(T(v), T(q), T(a))
        check(raw, fx.o["raw"], what="C1 raw, geometry %d" % cfg)
        pn = p.cpu().numpy()
        for b in range(4):
            for g in range(2):
                assert np.argmax(pn[b, ..., g]) == np.argmax(fx.o["p"][b, ..., g])


@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_forced_tile_geometry_and_multichunk_softmax_backward(cfg):
    fx = gu.load("g8_triattention_grad")
    m = _tri(fx)
    v, q, a = (T(fx.i[k]).requires_grad_(True) for k in ("v", "q", "a"))
    with cti_amd.ops.tuning(gemm_cfg=cfg, tri_chunk=16):
        p, logits = m(v, q, a)
        check(p, fx.o["p"], what="p under autograd, geometry %d" % cfg)
        (p * T(fx.i["cot_p"])).sum().backward()
    for n_, t_ in (("v", v), ("q", q), ("a", a)):
        check(t_.grad, fx.g[n_], what="d%s geometry %d" % (n_, cfg))
    check_param_grads(m, fx)


def test_f16f6_mode_on_the_small_fixtures_and_config_1():
    """precision='f16f6' (mode-3 product on f16 + block-scaled fp6 planes) against the reference fixtures: reduced dims (h = 64: the fused
    path with the MFMA M build; h = 48 is not a multiple of 32 and takes the bf16x3 kernels), an all-zero sample, BASELINE configs[0]."""
    old = cti_amd.get_precision()
    cti_amd.set_precision("f16f6")
    try:
        for name in ("g3_tcnet_small", "g3_tcnet_g3_odd", "g3_tcnet_allzero_sample"):
            fx = gu.load(name)
            m = _tri(fx)
            v, q, a = T(fx.i["v"]), T(fx.i["q"]), T(fx.i["a"])
            with torch.no_grad():
                raw = m.TriAtt(v, q, a)
                p, logits = m(v, q, a)
            check(raw, fx.o["raw"], what=name + " raw [f16f6]")
            assert np.array_equal(np.isneginf(logits.cpu().numpy()), np.isneginf(fx.o["logits"]))
            ok = ~np.isnan(fx.o["p"])
            assert np.max(np.abs(p.cpu().numpy()[ok] - fx.o["p"][ok])) < TOL * float(np.max(fx.o["p"][ok]))
        fx, params, v, q, a = gu.c1_case()
        m = _tri_from(fx.cfg, params)
        with torch.no_grad():
            raw = m.TriAtt(T(v), T(q), T(a))
            p, _ = m(T(v), T(q), T(a))
        e = check(raw, fx.o["raw"], what="C1 raw [f16f6]")
        e64 = check(raw, O.tcnet_forward(v, q, a, params, "TriAtt.", dtype=np.float64), what="C1 raw vs float64 [f16f6]")
        print("C1 [f16f6]: err vs reference %.3g, vs float64 truth %.3g" % (e, e64))
        pn = p.cpu().numpy()
        for b in range(4):
            for g in range(2):
                assert np.argmax(pn[b, ..., g]) == np.argmax(fx.o["p"][b, ..., g])
        # values far outside f16's range degrade gracefully (no NaN / inf): the hi part saturates, the fp6 lo part carries the rest at 4 bits
        big = cti_amd.ops.gemm_nt_f16f6(torch.full((40, 64), 3.0e5, device=DEV), torch.ones(24, 64, device=DEV))
        assert bool(torch.isfinite(big).all()) and float((big / (3.0e5 * 64) - 1).abs().max()) < 0.1
    finally:
        cti_amd.set_precision(old)


def test_softmax_partials_from_the_gemm_epilogue_match_the_two_pass_softmax():
    """precision='f16f6', glimpse 2: TriAttention's partial pass (per-wave max / sum of exponentials) comes out of the mode-3 GEMM's
    accumulators (cti_tcnet_forward_sm) and the softmax reads the logits once.  Same p and the same in-place -inf fill as the stand-alone
    masked softmax on the same logits: C2 widths with padded objects (10-36 real rows), an all-zero sample (NaN like the reference), and
    the reduced fixtures (A = 5 < one tile)."""
    old = cti_amd.get_precision()
    cti_amd.set_precision("f16f6")
    try:
        fx, params, v, q, a, idx = gu.c2_case()
        v = v.copy()
        v[1] = 0                                                   # an all-masked sample
        cases = [(_tri_from(fx.cfg, params), v, q, a)]
        for name in ("g3_tcnet_small", "g3_tcnet_allzero_sample"):
            f2 = gu.load(name)
            cases.append((_tri(f2), f2.i["v"], f2.i["q"], f2.i["a"]))
        used = 0
        for m, v_, q_, a_ in cases:
            with torch.no_grad():
                raw, mask, part = m.TriAtt(T(v_), T(q_), T(a_), _want_mask=True, _want_sm_partials=True)
                if part is None:
                    continue
                used += 1
                ref_logits = raw.clone()
                p_ref = cti_amd.ops.masked_softmax_tri_(ref_logits, mask)
                p = cti_amd.ops.masked_softmax_tri_from_partials_(raw, mask, part)
            pn, pr = p.cpu().numpy(), p_ref.cpu().numpy()
            assert np.array_equal(np.isnan(pn), np.isnan(pr))
            ok = ~np.isnan(pr)
            assert np.max(np.abs(pn[ok] - pr[ok])) <= 1e-5 * np.max(pr[ok])
            assert np.array_equal(raw.cpu().numpy(), ref_logits.cpu().numpy(), equal_nan=True)      # the same -inf fill, the same finite logits
            s = pn.reshape(pn.shape[0], -1, 2).sum(1)
            assert np.all(np.isnan(s) | (np.abs(s - 1) < 1e-4))
        assert used >= 1                                           # (the reduced fixtures have A = 5: the few-answer path, which leaves no partials)
    finally:
        cti_amd.set_precision(old)


@pytest.mark.parametrize("B,V,Q,A,vd,qd,ad,h,R", [
    (2, 44, 16, 9, 96, 64, 48, 64, 4),        # hr = 16, but X + hold buffer exceed the LDS: fp32 M + encoding pass; a side on f16f6 products
    (3, 10, 6, 7, 40, 24, 30, 32, 2),         # direct-encoding M build at its smallest; a_dim % 4 != 0 (the encoder's scalar loads); h = one K block
    (2, 12, 5, 40, 64, 32, 52, 64, 16),       # hr = 4: VALU M build + encoding pass
    (2, 36, 14, 200, 64, 48, 300, 96, 6),     # h = 96: a partial row tile of the transposed a-side products (feature blocks beyond h are skipped), K = 300 -> 10 blocks
])
def test_f16f6_forward_around_its_fast_paths(B, V, Q, A, vd, qd, ad, h, R):
    """precision='f16f6', fused TCNet.forward (src/tc.py:41-52) with more than 6 answer tokens at shapes on either side of every shape test
    of its a-side / M-build / mode-3 chain, against the float64 oracle: <= 1e-4 normalised (north_star)."""
    torch.manual_seed(B * 1000 + V + A)
    net = cti_amd.TCNet(vd, qd, ad, h, 1, R, 2).to(DEV).eval()
    g = torch.Generator().manual_seed(5)
    v = torch.randn(B, V, vd, generator=g).abs()
    v[0, V - 2:] = 0
    q = torch.tanh(torch.randn(B, Q, qd, generator=g))
    a = torch.tanh(torch.randn(B, A, ad, generator=g))
    params = {k: t.detach().cpu().numpy() for k, t in net.state_dict().items()}
    ref = O.tcnet_forward(v.numpy(), q.numpy(), a.numpy(), params, dtype=np.float64)
    old = cti_amd.get_precision()
    try:
        outs = {}
        for prec in ("bf16x3", "f16f6"):
            cti_amd.set_precision(prec)
            with torch.no_grad():
                outs[prec] = net(v.to(DEV), q.to(DEV), a.to(DEV)).cpu().numpy()
    finally:
        cti_amd.set_precision(old)
    e3, e6 = O.norm_max_err(outs["bf16x3"], ref), O.norm_max_err(outs["f16f6"], ref)
    print("TCNet.forward B=%d V=%d Q=%d A=%d h=%d R=%d: bf16x3 %.3g, f16f6 %.3g vs float64" % (B, V, Q, A, h, R, e3, e6))
    assert e3 < TOL and e6 < TOL
