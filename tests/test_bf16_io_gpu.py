"""bf16 activations of the plain-bf16 mode (round 5; BASELINE configs[2] / [3] name bf16 tensors): the image features `v` handed over as bf16, the hoisted
projection GEMMs writing bf16 rows (cti_gemm_bf16_rows, no split pass), and the consumers that stream those rows -- the shifted sum-pools
(reference src/bc.py:70-78, src/tc.py:54-61), the bilinear attention logits (src/bc.py:52-58), TriAttention's hoisted v side and the zero-row mask
(src/attention.py:36,55) -- reading them as they are.  Kernel level: bit-identical to the fp32-reading kernels on the widened values.  Model level: the
reference's BAN / CTI forwards at full widths against the oracle fed the bf16-rounded features.  Needs an MI355X."""
import os
import sys

import numpy as np
import pytest
import torch

import cti_amd
from oracle import cti_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ops = cti_amd.ops


@pytest.fixture(autouse=True)
def restore_precision():
    old = cti_amd.get_precision()
    yield
    cti_amd.set_precision(old)


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def test_zero_row_mask_of_bf16_rows_is_bit_exact():
    """mask[r] = every element of the row is +-0: -0, the smallest subnormal, NaN and inf in either half of a 32-bit word"""
    v = torch.zeros(9, 7, 64, dtype=torch.bfloat16)
    v[0, 1, 5] = -0.0
    v[1, 2, 6] = torch.tensor(9.18e-41).to(torch.bfloat16)          # a bf16 subnormal
    v[2, 3, 7] = float("nan")
    v[3, 4, 8] = float("inf")
    v[4, 5, 63] = 1.0
    v[5, 6, 0] = -1e-3
    v[6] = torch.randn(7, 64).to(torch.bfloat16)
    v[6, 2] = 0
    got = ops.zero_row_mask(v.to(DEV)).cpu()
    want = (v.float().abs() == 0).all(-1) & ~torch.isnan(v.float()).any(-1)
    assert got.dtype == torch.uint8 and torch.equal(got.bool(), want)
    assert torch.equal(got, ops.zero_row_mask(v.float().to(DEV)).cpu())


@pytest.mark.parametrize("prec", ["bf16", "bf16x3"])
@pytest.mark.parametrize("B,V,Q,D", [(256, 36, 14, 1024), (5, 9, 12, 256), (3, 50, 7, 64)])
def test_bi_pool_shift_reads_bf16_rows(prec, B, V, Q, D):
    cti_amd.set_precision(prec)
    vt = torch.relu(_rnd((B, V, D), 1)).to(torch.bfloat16)
    qt, qadd = _rnd((B, Q, D), 2), _rnd((B, D), 3, 0.3)
    w = torch.softmax(_rnd((B, V * Q), 4), 1).view(B, V, Q)
    a = ops.bi_pool_shift(vt, qt, qadd, w)
    b = ops.bi_pool_shift(vt.float(), qt, qadd, w)
    assert a is not None and torch.equal(a, b)
    ref = torch.einsum("bvd,bvq,bqd->bd", vt.double(), w.double(), torch.relu(qt.double() + qadd.double()[:, None, :]))
    assert O.norm_max_err(a.cpu().numpy(), ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("prec,tol", [("bf16", 1e-2), ("bf16x3", 2e-5)])
@pytest.mark.parametrize("B,V,Q,A,D,rep", [(256, 36, 14, 3, 1024, 1), (64, 36, 12, 6, 1024, 4), (6, 20, 9, 3, 64, 1)])
def test_tri_pool_shift_reads_bf16_rows(prec, tol, B, V, Q, A, D, rep):
    cti_amd.set_precision(prec)
    vt = torch.relu(_rnd((B // rep, V, D), 1)).to(torch.bfloat16)
    qt, at = _rnd((B, Q, D), 2), _rnd((B, A, D), 3)
    qadd, aadd = _rnd((B, D), 4, 0.3), _rnd((B, D), 5, 0.3)
    w = torch.softmax(_rnd((B, V * Q * A), 6), 1).view(B, V, Q, A)
    a = ops.tri_pool_shift(vt, qt, at, qadd, aadd, w, v_rep=rep)
    b = ops.tri_pool_shift(vt.float(), qt, at, qadd, aadd, w, v_rep=rep)
    assert a is not None and torch.equal(a, b)
    ref = torch.einsum("bvd,bvqa,bqd,bad->bd", vt.double().repeat_interleave(rep, 0), w.double(), torch.relu(qt.double() + qadd.double()[:, None, :]),
                       torch.relu(at.double() + aadd.double()[:, None, :]))
    assert O.norm_max_err(a.cpu().numpy(), ref.cpu().numpy()) < tol


@pytest.mark.parametrize("prec,tol", [("bf16", 1e-2), ("bf16x3", 3e-5)])
@pytest.mark.parametrize("B,G,V,Q,D", [(256, 8, 36, 14, 3072), (4, 2, 9, 7, 96), (3, 4, 50, 16, 128)])
def test_bi_logits_read_bf16_rows(prec, tol, B, G, V, Q, D):
    cti_amd.set_precision(prec)
    vt = torch.relu(_rnd((B, V, D), 1)).to(torch.bfloat16)
    qt, h = torch.relu(_rnd((B, Q, D), 2)), _rnd((G, D), 3, 0.1)
    hs, hb = torch.tensor([0.7], device=DEV), _rnd((G,), 4)
    a = ops.bi_logits(vt, qt, h, hs, hb)
    b = ops.bi_logits(vt.float(), qt, h, hs, hb)
    assert torch.equal(a, b)
    ref = 0.7 * torch.einsum("bvd,gd,bqd->bgvq", vt.double(), h.double(), qt.double()) + hb.double()[None, :, None, None]
    assert O.norm_max_err(a.cpu().numpy(), ref.cpu().numpy()) < tol


@pytest.mark.parametrize("config", ["c4", "c3"])
def test_full_models_take_bf16_image_features(config):
    """BASELINE configs[2] / [3] at full widths in the plain-bf16 mode with `v` handed over as bf16: no split pass over v, bf16 rows between the projection
    GEMMs and the pools / attention.  The logits agree with the same models fed the widened (fp32) copy of the same bf16 values -- the difference is the
    bf16 rounding of the projected v alone -- and with the oracle on the first rows at the mode's tolerance."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cti_amd.set_precision("bf16")
    torch.manual_seed(5)
    s = bench.model_setup(config, 256, 0, torch.device(DEV))
    assert s["v_bf16"]
    with torch.no_grad():
        out = s["fwd"]()
        torch.cuda.synchronize()
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for name, got, ref in s["oracle"](4, out):
            err = O.norm_max_err(got, ref)
            print(config, name, "vs oracle (bf16-rounded v): %.2e" % err)
            assert err < bench.MODEL_TOL["bf16"]
        # the same models on the same bf16-representable image features handed over as fp32 (the path of rounds 1-4: split pass, fp32 rows everywhere)
        i = s["inputs"]
        v32 = i["v"].float()
        if config == "c3":
            refs = (s["models"]["mc_cti"](v32, i["boxes"], i["q"], i["a"])[0],)
        else:
            refs = (s["models"]["ban"](v32, i["boxes"], i["q"], None)[0], s["models"]["cti"](v32, i["q"], i["a"]))
        torch.cuda.synchronize()
    for a, b in zip(outs, refs):
        worst = float(((a.float() - b.float()).abs().flatten(1).amax(1) / b.float().abs().max()).max())
        print(config, "every row, bf16 v vs fp32 v: %.2e" % worst)
        assert worst < bench.MODEL_TOL["bf16"]
