"""The two entry points the unrolled glimpse loops of round 5 stand on (base_model._ban_forward_unrolled / _tri_loop_unrolled; reference loops
src/FFOE/base_model.py:53-64, 129-134), each against float64 on its own:
  cti_bi_pool_shift_multi_fwd -- the shifted bi pool (src/bc.py:70-78 with q_net's ReLU formed on load) whose shift is a SUM of up to 32 addends, each a (B, ld)
                                 fp32 block or one row broadcast over the batch: the raw split-K slabs of the product that feeds it;
  cti_gemm_pb_partials        -- those raw slabs: x @ W^T against resident planes, left in their K ranges (no reduce pass, scale or bias).
Needs an MI355X."""
import numpy as np
import pytest
import torch

import cti_amd
from oracle import cti_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ops = cti_amd.ops
TOL = 2e-5


@pytest.fixture(autouse=True)
def restore_precision():
    old = cti_amd.get_precision()
    yield
    cti_amd.set_precision(old)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


@pytest.mark.parametrize("B,V,Q,D", [(5, 9, 7, 64), (4, 36, 14, 1024), (3, 36, 12, 512)])
@pytest.mark.parametrize("n_adds", [0, 1, 5, 32])
@pytest.mark.parametrize("vt16", [False, True])
def test_multi_addend_bi_pool_vs_float64(B, V, Q, D, n_adds, vt16):
    rs = np.random.RandomState(B * 1000 + n_adds * 10 + Q)
    vt = rs.standard_normal((B, V, D)).astype(np.float32)
    qt = rs.standard_normal((B, Q, D)).astype(np.float32)
    w = rs.rand(B, 3, V, Q).astype(np.float32)
    vt_dev = T(vt).to(torch.bfloat16) if vt16 else T(vt)
    vt_ref = vt_dev.float().cpu().numpy().astype(np.float64)
    # addends: every third one a single row for the whole batch (row stride 0), the others (B, ld) blocks with ld >= D (slices of a wider buffer)
    keep, adds, total = [], [], np.zeros((B, D), np.float64)
    for i in range(n_adds):
        if i % 3 == 2:
            a = rs.standard_normal((1, D)).astype(np.float32) / max(1, n_adds)
            t = T(a)
            adds.append((t.data_ptr(), 0))
            total += a.astype(np.float64)
        else:
            ld = D + 8 * (i % 2)
            a = rs.standard_normal((B, ld)).astype(np.float32) / max(1, n_adds)
            t = T(a)
            adds.append((t.data_ptr(), ld))
            total += a[:, :D].astype(np.float64)
        keep.append(t)
    wide = torch.full((B, 2 * D + 4), 7.0, device=DEV)                       # the pooled vectors of several glimpses side by side: a strided (B, D) view
    out = wide[:, D:2 * D]
    assert ops.bi_pool_shift_multi(vt_dev, T(qt), adds, T(w)[:, 1], out)
    q_ = np.maximum(qt.astype(np.float64) + total[:, None, :], 0)
    ref = np.einsum("bvd,bvq,bqd->bd", vt_ref, w[:, 1].astype(np.float64), q_)
    assert O.norm_max_err(out.cpu().numpy(), ref) < TOL
    assert float(wide[:, :D].min()) == 7.0 and float(wide[:, 2 * D:].min()) == 7.0      # nothing written beside the view
    # the single-addend entry point agrees bit for bit where both apply
    if n_adds == 1:
        one = ops.bi_pool_shift(vt_dev, T(qt), keep[0][:, :D].contiguous(), T(w)[:, 1])
        assert one is not None and torch.equal(one, out)


def test_multi_addend_bi_pool_declines_what_it_cannot_do():
    rs = np.random.RandomState(3)
    B, V, Q, D = 2, 9, 7, 64
    vt, qt, w = T(rs.standard_normal((B, V, D)).astype(np.float32)), T(rs.standard_normal((B, Q, D)).astype(np.float32)), T(rs.rand(B, V, Q).astype(np.float32))
    a = T(np.zeros((B, D), np.float32))
    out = torch.empty((B, D), device=DEV)
    assert ops.bi_pool_shift_multi(vt, qt, [(a.data_ptr(), D)] * 33, w, out) is False                       # more addends than a launch carries
    assert ops.bi_pool_shift_multi(vt[:, :, :D - 2].contiguous(), qt[:, :, :D - 2].contiguous(), [], w, out[:, :D - 2]) is False      # D % 4 != 0: no kernel with the on-load shift


@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("f16f6", 2e-5), ("bf16", 6e-3)])
@pytest.mark.parametrize("M,N,K,pad", [(256, 1024, 1024, 0), (256, 1024, 4096, 1024), (256, 1024, 7168, 1024), (5, 40, 96, 0), (33, 96, 544, 32)])
def test_raw_split_k_partials_sum_to_the_product(prec, tol, M, N, K, pad):
    """(S, M, N) slabs whose sum over S is x @ W^T; x may be the leading K columns of a wider row-major buffer (the pooled vectors of the glimpses so far)."""
    cti_amd.set_precision(prec)
    rs = np.random.RandomState(M + N + K)
    xw = rs.standard_normal((M, K + pad)).astype(np.float32)
    wt = rs.standard_normal((N, K)).astype(np.float32)
    x = T(xw)[:, :K]
    planes = ops.split_operand(T(wt))
    slabs = ops.gemm_pb_partials(x, planes, N)
    S = slabs.shape[0]
    assert slabs.shape == (S, M, N) and S >= 1 and S == ops.L.lib().cti_gemm_pb_partials_count(M, N, K)
    ref = xw[:, :K].astype(np.float64) @ wt.astype(np.float64).T
    assert O.norm_max_err(slabs.sum(0).cpu().numpy(), ref) < tol
    if K >= 4096 and M == 256:
        # round 6: batch-sized products leave ONE slab (K is split over the waves of cti_gemm_skinny.hip's workgroups and summed in LDS); with a forced tile
        # geometry -- the split-K path of rounds 1-5 -- the shapes of the unrolled loop really are split into slabs
        assert S == 1
        with ops.tuning(gemm_cfg=0):
            old = ops.gemm_pb_partials(x, planes, N)
            assert old.shape[0] > 1 and old.shape[0] == ops.L.lib().cti_gemm_pb_partials_count(M, N, K)
            assert O.norm_max_err(old.sum(0).cpu().numpy(), ref) < tol
    # the reduced product of the same operands agrees with the sum of the slabs
    full = ops.gemm_nt(x, T(wt), B_planes=planes)
    assert O.norm_max_err(slabs.sum(0).cpu().numpy(), full.cpu().numpy().astype(np.float64)) < tol
