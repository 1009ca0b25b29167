"""Range guard of the f16f6 forward (verdict r2 #2; reference src/Tensor.py:12,18 multiplies in full-range fp32): at the BASELINE configs[1]
WIDTHS (V=36x2048, Q=14x1024, A x 300, h_mm 512, rank 32, glimpse 2) the fused TCNet.forward in the f16f6 mode must return fp32-grade numbers
(<= 1e-4 normalised max error vs the float64 oracle) for inputs scaled x10^3 and x10^-4, with a 10^4 outlier in every 32-wide block, with
weight_g x100 and with NaN / inf inputs -- through the explicit bf16x3 fallback where the operand format's domain is left -- and never
silently clamped numbers: in the 'poison' mode (no host wait, what hipGraph capture forces) an out-of-range call returns NaN."""
import warnings

import numpy as np
import pytest
import torch

import cti_amd
import golden_util as gu
from oracle import cti_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4
ops = cti_amd.ops


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


@pytest.fixture(autouse=True)
def f16f6_mode():
    old = cti_amd.get_precision()
    cti_amd.set_precision("f16f6")
    ops._range_log.update(consecutive=0, skip=0); ops._range_owner.clear()
    yield
    cti_amd.set_precision(old)
    cti_amd.set_range_check("sync")
    ops._range_log.update(consecutive=0, skip=0); ops._range_owner.clear()


def _net(params):
    c = gu.load("g3_tcnet_forward_c2").cfg
    m = cti_amd.TCNet(c["v_dim"], c["q_dim"], c["a_dim"], c["h_dim"], 1, c["rank"], c["glimpse"])
    m.load_state_dict({k[len("TriAtt."):]: torch.from_numpy(np.asarray(v)) for k, v in params.items() if k.startswith("TriAtt.")})
    return m.to(DEV).eval()


def _case(A=640, B=2):
    fx, params, v, q, a, _ = gu.c2_case()
    return params, v[:B].copy(), q[:B].copy(), a[:B, :A].copy()


def _run(params, v, q, a):
    m = _net(params)
    before = ops.f16f6_range_status()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            out = m(T(v), T(q), T(a)).cpu().numpy()
    after = ops.f16f6_range_status()
    return out, after["calls"] - before["calls"], after["trips"] - before["trips"], after["last_status"]


def _err(out, ref):
    return float(np.max(np.abs(out - ref)) / np.max(np.abs(ref)))


def test_inputs_inside_the_domain_do_not_trip_the_guard():
    params, v, q, a = _case(A=3129)
    out, calls, trips, status = _run(params, v, q, a)
    assert (calls, trips, status) == (1, 0, 0)
    e = _err(out, O.tcnet_forward(v, q, a, params, "TriAtt.", dtype=np.float64))
    assert e < TOL, e
    print("f16f6 guard, clean inputs: err %.3g, no trip" % e)


@pytest.mark.parametrize("scale,bit", [(1e3, cti_amd.pkg._lib.GUARD_SATURATED), (1e-4, None), (1e-7, cti_amd.pkg._lib.GUARD_UNDERFLOW)])
def test_scaled_inputs_are_fp32_grade_or_take_the_explicit_bf16x3_fallback(scale, bit):
    """x10^3: M ~ V^ Q^ grows with the square, beyond f16's 65504 -- the guard must trip.  x10^-4: the layers' biases keep the intermediates in
    range and `a` itself (|x| ~ 1e-4, above the 2^-12 knee of its largest blocks) still encodes to 2^-15 of its block maxima: either verdict
    is fine as long as the numbers are right.  x10^-7: `a` lies entirely in f16's subnormal range -- the guard must trip.  In every case the
    caller gets fp32-grade numbers."""
    params, v, q, a = _case(A=3129)
    v, q, a = (v * np.float32(scale)), (q * np.float32(scale)), (a * np.float32(scale))
    out, calls, trips, status = _run(params, v, q, a)
    assert calls == 1
    if bit is not None:
        assert trips == 1 and (status & bit), (calls, trips, status)
    ref = O.tcnet_forward(v, q, a, params, "TriAtt.", dtype=np.float64)
    e = _err(out, ref)
    assert np.isfinite(out).all() and e < TOL, (e, trips, status)
    print("f16f6 guard, inputs x%g: trips %d status %d, err %.3g" % (scale, trips, status, e))


def test_one_large_outlier_per_block():
    """A 10^4 outlier in every 32-wide block of `a` sets that block's fp6 scales: <= 1e-4 directly, or through the fallback."""
    params, v, q, a = _case()
    a = a.copy()
    a.reshape(a.shape[0], a.shape[1], -1)[:, :, 5::32] = np.float32(1e4)
    out, calls, trips, status = _run(params, v, q, a)
    e = _err(out, O.tcnet_forward(v, q, a, params, "TriAtt.", dtype=np.float64))
    assert np.isfinite(out).all() and e < TOL, (e, trips, status)
    print("f16f6 guard, 1e4 outlier per block: trips %d status %d err %.3g" % (trips, status, e))


def test_trained_size_weight_gains():
    """weight_g x100 on every weight-normalised layer: V^, Q^ x10^4 -> M x10^8."""
    params, v, q, a = _case()
    params = {k: (np.asarray(x) * np.float32(100) if k.endswith("weight_g") else x) for k, x in params.items()}
    out, calls, trips, status = _run(params, v, q, a)
    ref = O.tcnet_forward(v, q, a, params, "TriAtt.", dtype=np.float64)
    e = _err(out, ref)
    assert trips == 1 and np.isfinite(out).all() and e < TOL, (e, trips, status)
    print("f16f6 guard, weight_g x100: status %d err %.3g (max |out| %.3g)" % (status, e, np.max(np.abs(ref))))


@pytest.mark.parametrize("which,bad", [("q", np.nan), ("a", np.inf), ("v", np.nan), ("a", np.nan)])
def test_non_finite_inputs_propagate_like_the_reference(which, bad):
    """The reference's fp32 matmuls propagate NaN / inf; the f16f6 encoders would map them to finite values.  The guard trips and the bf16x3
    re-run propagates: the NaN PATTERN equals the oracle's and the finite part is fp32-grade."""
    params, v, q, a = _case(A=64)
    x = {"v": v, "q": q, "a": a}[which]
    x[0, 1, 7] = bad
    out, calls, trips, status = _run(params, v, q, a)
    with np.errstate(all="ignore"):
        ref = O.tcnet_forward(v, q, a, params, "TriAtt.", dtype=np.float64)
    assert trips == 1 and status != 0
    assert not np.isfinite(out[0]).all() and np.isfinite(out[1]).all()          # loud in the sample that holds it, and only there
    if np.isnan(bad):
        assert np.array_equal(np.isfinite(out), np.isfinite(ref))               # the NaN pattern of the reference's arithmetic
    fin = np.isfinite(ref) & np.isfinite(out)
    assert np.max(np.abs(out[fin] - ref[fin])) / np.max(np.abs(ref[fin])) < TOL


def test_poison_mode_returns_nan_never_clamped_numbers():
    """range check 'poison' = no host wait (the form a hipGraph capture uses): an out-of-range call comes back all-NaN."""
    params, v, q, a = _case(A=64)
    cti_amd.set_range_check("poison")
    m = _net(params)
    with torch.no_grad():
        clean = m(T(v), T(q), T(a))
        bad = m(T(v * np.float32(1e3)), T(q * np.float32(1e3)), T(a * np.float32(1e3)))
        again = m(T(v), T(q), T(a))
    assert bool(torch.isfinite(clean).all()) and bool(torch.isnan(bad).all())
    assert torch.equal(clean, again)                               # the guard block is per call: a trip does not stick to later launches
    # ... and the guarded launch sequence is capturable: replay == eager, with the NaN fill inside the graph
    vs, qs, as_ = T(v), T(q), T(a)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        m(vs, qs, as_); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            o = m(vs, qs, as_)
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(o, clean)
    vs.mul_(1e3); qs.mul_(1e3); as_.mul_(1e3)
    g.replay(); torch.cuda.synchronize()
    assert bool(torch.isnan(o).all())


def test_repeated_trips_go_straight_to_bf16x3_for_a_while():
    params, v, q, a = _case(A=64)
    big = [T(x * np.float32(1e3)) for x in (v, q, a)]
    m = _net(params)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            outs = [m(*big) for _ in range(4)]
    st = ops.f16f6_range_status()
    assert st["skip"] > 0 and all(torch.equal(outs[0], o) for o in outs[1:])


def test_the_repeat_offender_shortcut_is_per_network():
    """two trips of ONE TCNet must not route every other network's calls to bf16x3 -- the counters are keyed by the network's own parameters."""
    params, v, q, a = _case(A=64)
    big = [T(x * np.float32(1e3)) for x in (v, q, a)]
    m, m2 = _net(params), _net(params)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            for _ in range(3):
                m(*big)
            assert ops.f16f6_range_status()["skip"] > 0
            calls0 = ops.f16f6_range_status()["calls"]
            o2 = m2(T(v), T(q), T(a))
            assert ops.f16f6_range_status()["calls"] == calls0 + 1 and ops.f16f6_range_status()["last_status"] == 0     # the guarded f16f6 form ran, in range
    assert bool(torch.isfinite(o2).all())
