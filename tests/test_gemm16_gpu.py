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


@pytest.mark.parametrize("M,N,K,nb,out_dtype", [
    (256, 256, 128, 1, torch.float32), (1000, 776, 320, 1, torch.float32), (513, 260, 160, 1, torch.bfloat16), (700, 520, 256, 3, torch.bfloat16),
    (9216, 3072, 512, 1, torch.float32),
    (9216, 3072, 256, 1, torch.bfloat16), (2305, 1024, 2048, 1, torch.bfloat16),
])
def test_row_major_bf16_operand_and_bf16_rows_output(M, N, K, nb, out_dtype):
    """cti_gemm_bf16_rows: A read as a row-major bf16 matrix (no split pass), fp32 or bf16 rows out; both tile geometries"""
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = bf16r(torch.randn(nb * N, K, generator=g) / 4)
    b = torch.randn(nb * N, generator=g)
    wp = ops.split_operand(w.to(DEV), prec="bf16")
    y = ops.gemm_bf16_rows(a.to(DEV), wp, nb * N, nb1=nb, rA1=0, rB1=N, M=M, N=N, out_dtype=out_dtype, bias=b.to(DEV), bias_bs=N, relu=True)
    assert y.dtype == out_dtype
    ref = torch.relu((a.double() @ w.double().t() + b.double()).view(M, nb, N).permute(1, 0, 2))
    err = float((y.double().cpu().view(nb, M, N) - ref).abs().max() / ref.abs().max())
    print("gemm_bf16_rows %dx%dx%d nb=%d %s: %.2e" % (M, N, K, nb, out_dtype, err))
    assert err < (3e-6 if out_dtype == torch.float32 else 4e-3), err       # bf16 rows: one rounding of the result to 8 significant bits


@pytest.mark.parametrize("M,N,K,nb,out_dtype", [
    (4608, 4096, 512, 1, torch.float32),          # 288 tiles on 256 compute units, 16 stages per tile: ranges of 18 stages (heads / tails of 2 .. 16)
    (9216, 3072, 2048, 1, torch.bfloat16),        # configs[2]'s hoisted projection: 432 tiles, ranges of 108 stages
    (9216, 1024, 1024, 3, torch.float32),         # batched: 432 tiles over three weight sets, 32 stages per tile
    (9000, 3000, 768, 1, torch.float32),          # ragged edges: 36 x 12 tiles with partial last rows / columns
    (5120, 4352, 1056, 1, torch.bfloat16),        # 340 tiles, 33 stages: 256 does not divide the stage count (ranges of 43 and 44)
    (9216, 8192, 1024, 1, torch.bfloat16),        # configs[3]'s hoisted projections: 1 152 tiles = 4.5 rounds, the shape the default cuts (three whole rounds in front)
])
def test_stream_k_is_bit_identical_to_whole_tiles(M, N, K, nb, out_dtype):
    """cti_gemm_bf16_rows_sk (round 6): tiles cut along K, the second contributor starting from the first one's accumulators -- the same additions in the
    same order as the uncut product, so EVERY bit agrees with cti_gemm_bf16_rows; and against float64; flags back at zero, no error word; twice (the
    second launch finds the workspace the first one left)"""
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the shapes are chosen for 256 compute units")
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = bf16r(torch.randn(nb * N, K, generator=g) / 4)
    b = torch.randn(nb * N, generator=g).to(DEV)
    wp = ops.split_operand(w.to(DEV), prec="bf16")
    kw = dict(nb1=nb, rA1=0, rB1=N, M=M, N=N, out_dtype=out_dtype, bias=b, bias_bs=N, relu=True)
    y0 = ops.gemm_bf16_rows(a, wp, nb * N, stream_k=False, **kw)
    for rep in range(2):
        with ops.tuning(gemm16_sk=1):                 # (the default cuts only products of three or more rounds of tiles)
            y1 = ops.gemm_bf16_rows(a, wp, nb * N, stream_k=True, **kw)
        assert torch.equal(y0, y1), (rep, float((y0.float() - y1.float()).abs().max()))
        assert ops.gemm16_sk_state() == (0, 0)
    cols = torch.arange(0, N, 7)
    ref = torch.relu((a.cpu().double() @ w.double().t() + b.cpu().double()).view(M, nb, N).permute(1, 0, 2))[..., cols]
    err = float((y1.double().cpu().view(nb, M, N)[..., cols] - ref).abs().max() / ref.abs().max())
    assert err < (3e-6 if out_dtype == torch.float32 else 4e-3), err


def test_stream_k_on_two_streams_at_once():
    """two cut products in flight on sibling streams (each stream has its own workspace): both bit-identical to the uncut product, ten rounds"""
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the shape is chosen for 256 compute units")
    g = torch.Generator().manual_seed(5)
    M, N, K = 4608, 4096, 1024
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    wp = ops.split_operand(bf16r(torch.randn(N, K, generator=g) / 4).to(DEV), prec="bf16")
    y0 = ops.gemm_bf16_rows(a, wp, N, stream_k=False)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(10):
        for s in (s1, s2):
            with torch.cuda.stream(s), ops.tuning(gemm16_sk=1):
                outs.append(ops.gemm_bf16_rows(a, wp, N))
    torch.cuda.synchronize()
    assert all(torch.equal(y0, y) for y in outs)
    assert ops.gemm16_sk_state() == (0, 0)


def test_wide_tile_geometry_in_a_child_process():
    """CTI_GEMM16_TILE=1 (288 x 192 tiles; read once per process): every entry of two products against float64"""
    import os, subprocess, sys
    code = r"""
import torch, cti_amd
ops = cti_amd.ops
g = torch.Generator().manual_seed(3)
for M, N, K in ((1000, 776, 320), (9216, 3072, 256)):
    a = torch.randn(M, K, generator=g).to(torch.bfloat16); w = (torch.randn(N, K, generator=g) / 4).to(torch.bfloat16).float(); b = torch.randn(N, generator=g)
    y = ops.gemm_bf16_rows(a.cuda(), ops.split_operand(w.cuda(), prec='bf16'), N, bias=b.cuda(), relu=True)
    ref = torch.relu(a.double() @ w.double().t() + b.double())
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 3e-6, (M, N, K, err)
print('wide ok')
"""
    env = dict(os.environ, CTI_GEMM16_TILE="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "wide ok" in r.stdout, r.stdout[-2000:]


def test_bf16_rows_entry_refuses_what_the_kernel_cannot_read():
    a = torch.randn(64, 40, device=DEV).to(torch.bfloat16)                  # K = 40: not whole 64-B stages
    wp = ops.split_operand(torch.randn(32, 40, device=DEV), prec="bf16")
    with pytest.raises(Exception):
        ops.gemm_bf16_rows(a, wp, 32)


@pytest.mark.parametrize("M,N,K", [(256, 1024, 1024), (256, 3129, 2048), (3584, 1024, 1024), (130, 520, 328), (256, 1024, 600)])
def test_fp32_rows_read_directly_by_the_plain_bf16_product(M, N, K):
    """CTI_AF32_PB=1 (experiment, off by default: measured slower): the batch-sized products of the model forwards (cti_gemm_nt_pb below 128 tiles of
    256 x 256) take their fp32 activations as they stand -- no split launch -- also under split-K (a K range per workgroup offsets the fp32 rows); every
    entry against float64 on the bf16-rounded operands.  The knob is read once per process: a child process."""
    import os, subprocess, sys
    if os.environ.get("CTI_AF32_PB") != "1":
        env = dict(os.environ, CTI_AF32_PB="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", __file__, "-k", "test_fp32_rows_read_directly and %d-%d-%d" % (M, N, K)], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:]
        return
    g = torch.Generator().manual_seed(M + N + K)
    a = bf16r(torch.randn(M, K, generator=g)); w = bf16r(torch.randn(N, K, generator=g) / 8); b = torch.randn(N, generator=g)
    wp = ops.split_operand(w.to(DEV), prec="bf16")
    y = ops.gemm_nt(a.to(DEV), w.to(DEV), prec="bf16", B_planes=wp, bias=b.to(DEV), relu=True)
    ref = torch.relu(a.double() @ w.double().t() + b.double())
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 3e-6, err


@pytest.mark.parametrize("M,nb,N,K", [(256, 1, 2048, 1024), (256, 2, 1024, 2048), (256, 1, 3129, 2048), (100, 1, 72, 256), (33, 3, 16, 288), (512, 1, 1024, 1024), (256, 1, 1024, 8192)])
@pytest.mark.parametrize("mode,tol", [("bf16x3", 2e-5), ("bf16", 1e-2)])
def test_skinny_products_in_one_launch(M, nb, N, K, mode, tol):
    """Round 6: the products a caller plans a split-K for (few rows, long K: the classifier and residual layers at the batch's 256 rows) run as ONE launch of
    cti_gemm_skinny.hip -- K split over a workgroup's eight waves, A from the fp32 rows or from planes, partial tiles summed in LDS in wave order -- against float64,
    with scale / bias / ReLU, batches, ragged M and N, in both arithmetic modes; CTI_TUNE_GEMM_CFG keeps the split-K path reachable (same result within rounding);
    the same bits on every run."""
    import cti_amd
    ops = cti_amd.pkg.ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(nb * N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(nb * N, generator=g).cuda()
    scale = (torch.rand(nb, generator=g) + 0.5).cuda()
    old = cti_amd.get_precision()
    try:
        cti_amd.set_precision(mode)
        wp = ops.split_operand(w)
        ref = torch.relu((a.double() @ w.double().t()).view(M, nb, N) * scale.double().view(1, nb, 1) + bias.double().view(1, nb, N)).permute(1, 0, 2)

        def run():
            return ops.gemm_nt(a, w, nb1=nb, rA1=0, rB1=N, M=M, N=N, scale=scale, scale_div=N, scale_bs=1, bias=bias, bias_bs=N, relu=True, B_planes=wp)
        out = run()
        assert float((out.double().reshape(ref.shape) - ref).abs().max() / ref.abs().max()) < tol
        assert torch.equal(out, run())
        with ops.tuning(gemm_cfg=0):                                    # the split-K path of rounds 1-5
            old_path = run()
        assert float((out.double() - old_path.double()).abs().max() / ref.abs().max()) < tol
    finally:
        cti_amd.set_precision(old)
