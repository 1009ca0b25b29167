"""Accuracy envelope of the f16f6 default INSIDE its range (the reference multiplies in fp32, src/Tensor.py:12-20): the product's error is
~2^-17 sum_k |M_k A^_k| per output, i.e. 1e-4 of the LARGEST output only while the mode-3 contraction does not cancel too heavily.  At the BASELINE
configs[1] WIDTHS against the float64 oracle: (a) T_g with a common offset, (b) an operand pair built to cancel (max |out| << sum |M| |A^|),
(c) a trained-like state (weight gains x3, biases x10, half the ReLUs dead) -- each either within 1e-4 as it stands or tripped by the guard's
cancellation estimate and re-run in a mode that is.  Needs an MI355X."""
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
    ops._range_debug = True
    yield
    ops._range_debug = False
    cti_amd.set_precision(old)
    cti_amd.set_range_check("sync")
    ops._range_log.update(consecutive=0, skip=0); ops._range_owner.clear()


def _case(A=640, B=2):
    fx, params, v, q, a, _ = gu.c2_case()
    return {k[len("TriAtt."):]: np.asarray(x).copy() for k, x in params.items() if k.startswith("TriAtt.")}, v[:B].copy(), q[:B].copy(), a[:B, :A].copy()


def _run(sd, v, q, a):
    c = gu.load("g3_tcnet_forward_c2").cfg
    m = cti_amd.TCNet(c["v_dim"], c["q_dim"], c["a_dim"], c["h_dim"], 1, c["rank"], c["glimpse"])
    m.load_state_dict({k: torch.from_numpy(x) for k, x in sd.items()})
    m = m.to(DEV).eval()
    before = ops.f16f6_range_status()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            out = m(T(v), T(q), T(a)).cpu().numpy()
    after = ops.f16f6_range_status()
    ref = O.tcnet_forward(v, q, a, sd, dtype=np.float64)
    err = float(np.max(np.abs(out - ref)) / np.max(np.abs(ref)))
    return err, after["trips"] - before["trips"], after["last_status"], after["last_ratio"]


def test_synthetic_inputs_sit_well_inside_the_envelope():
    sd, v, q, a = _case()
    err, trips, status, rho = _run(sd, v, q, a)
    print("benign: err %.2e rho %.1f" % (err, rho))
    assert trips == 0 and status == 0 and err < TOL
    assert 0.5 < rho < 5.0                                    # the trip threshold (10) keeps a margin over the benign estimate


def test_core_tensor_with_a_common_offset():
    """(a) T_g + 8: every M entry carries a large common part; A^ >= 0 (ReLU), so the sum does not cancel -- the outputs grow with it"""
    sd, v, q, a = _case()
    sd["T_g"] = sd["T_g"] + 8.0
    err, trips, status, rho = _run(sd, v, q, a)
    print("T_g + 8: err %.2e rho %.1f trips %d status %d" % (err, rho, trips, status))
    assert err < TOL


def test_trained_like_state():
    """(c) weight gains x3, biases x10 with a negative shift that kills about half of the ReLUs"""
    sd, v, q, a = _case()
    rs = np.random.RandomState(3)
    for k in list(sd):
        if k.endswith("weight_g"):
            sd[k] = np.asarray(sd[k] * 3.0, dtype=np.float32)
        elif k.endswith("bias"):
            sd[k] = (sd[k] * 10.0 - 0.5 * np.abs(rs.standard_normal(sd[k].shape))).astype(np.float32)
    err, trips, status, rho = _run(sd, v, q, a)
    print("trained-like: err %.2e rho %.1f trips %d status %d" % (err, rho, trips, status))
    assert err < TOL


@pytest.mark.parametrize("eps,expect", [(1.0, 0), (0.25, 0), (0.13, 8), (0.05, 24), (0.004, 24)])
def test_cancelling_contraction_trips_the_estimate_and_is_rerun(eps, expect):
    """(b) the rank nets' LAST layer of the answer side built so that A^ is (nearly) orthogonal to every row of M: columns come in equal pairs and
    T_g's matching k slices in opposite pairs, plus eps of the original signal.  max |out| shrinks with eps while sum |M| |A^| stays: the estimate
    must trip (bf16x3 at moderate, exact fp32 at heavy cancellation) before the error passes 1e-4."""
    sd, v, q, a = _case()
    R = 32
    hr = sd["T_g"].shape[2]
    # answer side: rank net r, output k and k ^ 1 share their weights and bias -> A^[a, r, k] == A^[a, r, k ^ 1]
    for r in range(R):
        for nm in ("weight_v", "bias"):
            w = sd["a_net.%d.main.1.%s" % (r, nm)]
            w[1::2] = w[0::2]
    # core: in EFFECTIVE coordinates (the view scramble of src/Tensor.py:6-8, oracle teff_index_map) T_eff[.., k ^ 1, ..] = -T_eff[.., k, ..] on the
    # cancelling part, eps of the original on top
    G = sd["T_g"].shape[5]
    imap = O.teff_index_map(hr, G).reshape(-1)                      # flat index into T (hr, hr, hr, G) of the element acting at (i, j, k, g)
    Tg = sd["T_g"].copy()
    for r in range(R):
        flat = Tg[0, r, :, :, :, :, 0].reshape(-1)                  # a copy
        eff = flat[imap].reshape(hr, hr, hr, G)
        can = eff.copy()
        can[:, :, 1::2, :] = -can[:, :, 0::2, :]
        new = (can + eps * eff).reshape(-1)
        out_flat = np.empty_like(flat)
        out_flat[imap] = new
        Tg[0, r, :, :, :, :, 0] = out_flat.reshape(hr, hr, hr, G)
    sd["T_g"] = Tg.astype(np.float32)
    err, trips, status, rho = _run(sd, v, q, a)
    print("cancelling eps=%g: err %.2e rho %.1f trips %d status %d" % (eps, err, rho, trips, status))
    assert err < TOL
    if expect:
        assert trips == 1 and (status & 24) == expect, (status, expect)
    else:
        assert trips == 0
