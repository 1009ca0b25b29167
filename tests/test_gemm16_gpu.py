"""The round-4 plain-bf16 GEMM (csrc/cti_gemm16.hip: 256 x 256 tile, two wave groups one interval apart; it serves FCNet.forward of
reference src/fc.py:33-34 in the precision='bf16' mode) against float64 on bf16-ROUNDED operands, EVERY entry: with the operands rounded the
only error left is the fp32 accumulation order, so a wrong tile / wave / lane mapping, a missed ring wait or a lost epilogue constant shows as an
O(1) error somewhere.  Ragged edges, batches, split-K, scale / bias / ReLU, and the planes epilogue through the fused TCNet.forward.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import cti_amd
from oracle import cti_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ops = cti_amd.ops


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("M,N,K,nb,scale,bias,relu", [
    (256, 256, 128, 1, False, False, False),          # one tile, four stages
    (256, 256, 32, 1, False, False, False),           # one stage: the ring's prologue / tail only
    (1000, 777, 320, 1, True, True, True),            # ragged rows and columns, K padded to 320
    (300, 3129, 512, 1, False, True, False),          # the mode-3 product's column count
    (513, 257, 96, 1, True, False, True),             # a tile with one row / one column
    (700, 520, 256, 3, True, True, True),             # batches (nb1) with per-batch bias
    (2304, 1024, 2048, 1, True, True, True),          # the hoisted projections' depth: 64 stages per tile, several tiles per workgroup at 36 tiles
    (9216, 3072, 512, 1, False, True, True),          # 432 tiles on 256 workgroups: the persistent stream across tile boundaries
])
def test_every_entry_against_float64_of_the_rounded_operands(M, N, K, nb, scale, bias, relu):
    g = torch.Generator().manual_seed(M * 7 + N + K + nb)
    a = bf16r(torch.randn(M, K, generator=g))
    w = bf16r(torch.randn(nb * N, K, generator=g) / 4)
    b = torch.randn(nb * N, generator=g) if bias else None
    s = (torch.rand(nb * N // 8 + 1, generator=g) + 0.5) if scale else None
    with ops.tuning(gemm_cfg=2):
        wp = ops.split_operand(w.to(DEV), prec="bf16")
        y = ops.gemm_nt(a.to(DEV), w.to(DEV), nb1=nb, rA1=0, rB1=N, M=M, N=N, prec="bf16", B_planes=wp,
                        scale=None if s is None else s.to(DEV), scale_div=8, scale_bs=N // 8 if s is not None else 0,
                        bias=None if b is None else b.to(DEV), bias_bs=N if b is not None else 0, relu=relu)
    ref = (a.double() @ w.double().t()).view(M, nb, N).permute(1, 0, 2)
    if s is not None:
        cols = torch.arange(N)
        sc = torch.stack([s[z * (N // 8) + cols // 8] for z in range(nb)]).double()       # scale[b1 * scale_bs + n / scale_div]
        ref = ref * sc[:, None, :]
    if b is not None:
        ref = ref + b.double().view(nb, 1, N)
    if relu:
        ref = torch.relu(ref)
    err = float((y.double().cpu().view(nb, M, N) - ref).abs().max() / ref.abs().max())
    print("gemm16 %dx%dx%d nb=%d: %.2e" % (M, N, K, nb, err))
    assert err < 3e-6, err


def test_split_k_partials_through_the_new_kernel():
    """batch-sized M: plan_ksplit turns K ranges into extra workgroups (nb2 > 1, kc2 chunk offsets)"""
    g = torch.Generator().manual_seed(5)
    a = bf16r(torch.randn(256, 3072, generator=g)); w = bf16r(torch.randn(1024, 3072, generator=g) / 8); b = torch.randn(1024, generator=g)
    with ops.tuning(gemm_cfg=2):
        y = ops.gemm_nt(a.to(DEV), w.to(DEV), prec="bf16", B_planes=ops.split_operand(w.to(DEV), prec="bf16"), bias=b.to(DEV), relu=True)
    ref = torch.relu(a.double() @ w.double().t() + b.double())
    assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 3e-6


def test_planes_epilogue_through_the_fused_forward():
    """precision='bf16' TCNet.forward with the 256 x 256 tile forced: the Tucker projections leave as chunk-major planes (epilogue 1) that the rank
    nets' GEMM reads; against the float64 oracle at the plain-bf16 mode's tolerance, and against the same call on the default tiles to fp32 grade."""
    torch.manual_seed(21)
    net = cti_amd.TCNet(96, 64, 48, 64, 1, 4, 2).to(DEV).eval()
    rs = np.random.RandomState(7)
    v = np.abs(rs.standard_normal((5, 40, 96))).astype(np.float32); q = rs.standard_normal((5, 14, 64)).astype(np.float32)
    a = rs.standard_normal((5, 70, 48)).astype(np.float32)
    sd = {k: t.detach().cpu().numpy() for k, t in net.state_dict().items()}
    ref = O.tcnet_forward(v, q, a, sd, dtype=np.float64)
    T = lambda x: torch.from_numpy(x).to(DEV)      # noqa: E731
    old = cti_amd.get_precision()
    try:
        cti_amd.set_precision("bf16")
        with torch.no_grad():
            y_default = net(T(v), T(q), T(a)).cpu().numpy()
            with ops.tuning(gemm_cfg=2):
                net._prep_key = None
                y_big = net(T(v), T(q), T(a)).cpu().numpy()
    finally:
        cti_amd.set_precision(old)
        net._prep_key = None
    nrm = np.abs(ref).max()
    assert np.abs(y_big - ref).max() / nrm < 2e-2
    assert np.abs(y_big - y_default).max() / nrm < 1e-5
