"""The f16 + block-scaled-fp6 split product (csrc/cti_f16f6.h, cti_gemm_f16f6.hip): encoder bit-exact against a numpy restatement of the
format, GEMM against float64 on random and adversarial operands.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import cti_amd

pytestmark = pytest.mark.gpu
DEV = "cuda"
ops = cti_amd.ops


def e2m3_grid():
    idx = np.arange(32)
    return np.where(idx < 16, idx / 8.0, np.where(idx < 24, 2 + (idx - 16) / 4.0, 4 + (idx - 24) / 2.0))


def np_encode(x):
    """numpy restatement of the encoder (cti_f16f6.h) for (rows, K) float32, K % 32 == 0 -> (h16, codes_hi, codes_lo, sh, sl), codes as uint8
    per element.  The hi codes are what the GEMM derives in registers; only their scale byte is stored."""
    rows, K = x.shape
    h = np.clip(x, -65504, 65504).astype(np.float16)
    hf = h.astype(np.float32)
    lo = x - hf

    def enc(v):
        vb = v.reshape(rows, K // 32, 32)
        m = np.abs(vb).max(-1)
        u = m.astype(np.float32).view(np.uint32)
        E = ((u >> 23) & 0xff).astype(np.int64)
        byte = np.clip(E - np.where((u & 0x7fffff) > 0x700000, 1, 2), 1, 254)
        inv = ((254 - byte).astype(np.uint32) << 23).view(np.float32)
        y = vb * inv[..., None]
        ay = np.abs(y)
        idx = np.where(ay < 2, np.rint(ay * 8), np.where(ay < 4, 16 + np.rint((ay - 2) * 4), np.minimum(24 + np.rint((ay - 4) * 2), 31)))
        code = idx.astype(np.uint8) | (np.signbit(y).astype(np.uint8) << 5)
        return code.reshape(rows, K), byte.astype(np.uint8)
    ch, sh = enc(hf)
    cl, sl = enc(lo)
    return h, ch, cl, sh, sl


# position p of a block's 24 lo-code bytes holds element PI[p] (cti_f16f6.h f6_pi: the order the GEMM's lanes meet the block in)
PI = np.array([p + 8 if 8 <= p < 16 else (p - 8 if 16 <= p < 24 else p) for p in range(32)])


def unpack(block, rows, K, batch_rows=0):
    """The plane block -> (h16 (rows,K), codes_lo (rows,K) uint8 in ELEMENT order, sh, sl (rows,Kb)) following cti_f16f6.h's carve."""
    Kb = (K + 31) // 32
    r8 = lambda v: (v + 7) // 8 * 8                                  # noqa: E731
    pr = rows if not batch_rows else (rows + batch_rows - 1) // batch_rows * r8(batch_rows)
    ra, rs = r8(pr + 256), r8(pr + 512)
    b = block.cpu().numpy()
    off = 0

    def take(n):
        nonlocal off
        v = b[off:off + n]
        off = (off + n + 255) // 256 * 256
        return v
    H = take(Kb * ra * 64).view(np.float16).reshape(Kb, ra, 32)
    FL = take(Kb * ra * 24).reshape(Kb, ra, 24)
    S = take(Kb * rs * 2).reshape(Kb, rs, 2)
    prow = np.arange(rows) if not batch_rows else (np.arange(rows) // batch_rows) * r8(batch_rows) + np.arange(rows) % batch_rows

    def codes(F):
        Fd = F[:, prow, :].reshape(Kb, rows, 6, 4)[:, :, [0, 1, 4, 2, 3, 5], :].reshape(Kb, rows, 24)     # the six dwords are stored in the order [0 1 3 4 2 5]
        bits = np.unpackbits(Fd, axis=-1, bitorder="little").reshape(Kb, rows, 32, 6)
        by_position = (bits * (1 << np.arange(6))).sum(-1).astype(np.uint8)
        return by_position[:, :, PI].transpose(1, 0, 2).reshape(rows, Kb * 32)       # PI is an involution: element k sits at position PI[k]
    return (H[:, prow, :].transpose(1, 0, 2).reshape(rows, Kb * 32), codes(FL), S[:, prow, 0].T, S[:, prow, 1].T)


@pytest.mark.parametrize("rows,K,batch", [(70, 64, 0), (37, 96, 0), (45, 64, 9), (300, 512, 0), (5, 40, 0)])
def test_encoder_is_bit_exact(rows, K, batch):
    g = torch.Generator().manual_seed(rows * 1000 + K)
    x = torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, 1, generator=g) * 3)
    x[0, :5] = torch.tensor([0.0, -0.0, 7e4, -1e-7, 1.0])
    blk = ops.quantize_f16f6(x.to(DEV), batch)
    Kp = (K + 31) // 32 * 32
    xp = np.zeros((rows, Kp), np.float32)
    xp[:, :K] = x.numpy()
    h, ch, cl, sh, sl = np_encode(xp)
    H, CL, SH, SL = unpack(blk, rows, K, batch)
    assert np.array_equal(H.view(np.uint16), h.view(np.uint16))
    assert np.array_equal(SH, sh) and np.array_equal(SL, sl)
    assert np.array_equal(CL, cl)
    # the decoded value is within the format's promise of the input (hi + lo reconstruction, |x| <= 65504)
    grid = e2m3_grid()
    dec = lambda c, s: np.where(c & 32, -1.0, 1.0) * grid[c & 31] * np.repeat(2.0 ** (s.astype(np.float64) - 127), 32, axis=1)   # noqa: E731
    rec = h.astype(np.float64) + dec(cl, sl)
    # hi + decoded lo reconstructs x to 2^-15 of its BLOCK's largest magnitude (the lo codes carry 4 bits below the f16 residual of the
    # block maximum); saturated elements (|x| > 65504) excluded
    bmax = np.repeat(np.abs(xp).reshape(rows, -1, 32).max(-1), 32, axis=1)
    ok = np.abs(xp) <= 65504
    assert np.max((np.abs(rec - xp) / np.maximum(bmax, 1e-30))[ok & (bmax <= 65504)]) < 2.0 ** -15


def _ref(A, B, nb, M, N, gdiv):
    A64, B64 = A.double().view(nb, M, -1), B.double().view(nb, N, -1)
    C = torch.einsum("zmk,znk->zmn", A64, B64)
    if gdiv > 1:
        C = C.view(nb, M // gdiv, gdiv, N).permute(0, 1, 3, 2).contiguous()
    return C.numpy()


@pytest.mark.parametrize("nb,M,N,K,gdiv", [(1, 300, 200, 64, 1), (1, 256, 192, 512, 1), (3, 40, 29, 96, 2), (2, 1008, 3129, 512, 2), (1, 7, 5, 32, 1),
                                           (5, 130, 391, 160, 1), (1, 1000, 600, 1024, 1)])
def test_gemm_against_float64(nb, M, N, K, gdiv):
    g = torch.Generator().manual_seed(nb * 7 + M + N + K)
    A = torch.randn(nb * M, K, generator=g) * 3.0
    B = torch.relu(torch.randn(nb * N, K, generator=g))            # half zeros, like the rank nets' ReLU outputs
    C = ops.gemm_nt_f16f6(A.to(DEV), B.to(DEV), nb=nb, M=M, N=N, gdiv=gdiv).cpu().numpy()
    ref = _ref(A, B, nb, M, N, gdiv)
    err = np.abs(C - ref).max() / np.abs(ref).max()
    print("f16f6 GEMM nb=%d %dx%dx%d gdiv=%d: normalised max error %.3g" % (nb, M, N, K, gdiv, err))
    assert err < 5e-5


def test_gemm_epilogue_scale_bias_relu_and_dynamic_range():
    g = torch.Generator().manual_seed(3)
    M, N, K = 100, 70, 128
    A = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g) * 2)      # rows of very different magnitude
    B = torch.randn(N, K, generator=g) * torch.exp(torch.randn(1, K, generator=g))          # columns too
    scale = torch.rand(N // 10, generator=g) + 0.5
    bias = torch.randn(N, generator=g)
    C = ops.gemm_nt_f16f6(A.to(DEV), B.to(DEV), scale=scale.to(DEV), scale_div=10, bias=bias.to(DEV), relu=True).cpu().numpy()[0]
    ref = torch.relu((A.double() @ B.double().T) * scale.double().repeat_interleave(10)[None, :] + bias.double()[None, :]).numpy()
    rowmax = (A.double() @ B.double().T).abs().max(1, keepdim=True).values.numpy()
    assert np.max(np.abs(C - ref) / rowmax) < 1e-4                                             # per-row normalisation: every row keeps its accuracy


@pytest.mark.parametrize("seed", range(12))
def test_gemm_random_shapes_cover_the_stream_protocol(seed):
    """Randomised shapes around the kernel's protocol edges: one K block per tile (K = 32), fewer blocks than ring slots, several tiles per
    workgroup with 1-3 blocks each (total tiles > 256 CUs: the continuous stream crosses tile boundaries every few barriers), partial edge
    tiles in both dimensions, batch strides that are not multiples of the tile."""
    rs = np.random.RandomState(1000 + seed)
    K = int(rs.choice([32, 64, 96, 128, 160]))
    gdiv = int(rs.choice([1, 2]))
    if seed % 3 == 0:                                              # many small tiles: > 256 tiles, several per workgroup
        nb, M, N = int(rs.randint(40, 90)), int(rs.randint(130, 520)) // gdiv * gdiv, int(rs.randint(100, 400))
    elif seed % 3 == 1:                                            # one batch, odd sizes
        nb, M, N = 1, int(rs.randint(1, 1500)) // gdiv * gdiv + gdiv, int(rs.randint(1, 900))
    else:
        nb, M, N = int(rs.randint(2, 9)), int(rs.randint(200, 700)) // gdiv * gdiv, int(rs.randint(150, 650))
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(nb * M, K, generator=g) * 2.0
    B = torch.randn(nb * N, K, generator=g)
    C = ops.gemm_nt_f16f6(A.to(DEV), B.to(DEV), nb=nb, M=M, N=N, gdiv=gdiv).cpu().numpy()
    ref = _ref(A, B, nb, M, N, gdiv)
    err = np.abs(C - ref).max() / np.abs(ref).max()
    assert err < 5e-5, "nb=%d M=%d N=%d K=%d gdiv=%d: %.3g" % (nb, M, N, K, gdiv, err)


def _decode(H, CL, SL):
    grid = e2m3_grid()
    return H.astype(np.float64) + np.where(CL & 32, -1.0, 1.0) * grid[CL & 31] * np.repeat(2.0 ** (SL.astype(np.float64) - 127), 32, axis=1)


@pytest.mark.parametrize("rows,M,K,batch", [(500, 64, 64, 0), (192 * 3, 512, 512, 0), (1000, 96, 160, 125), (37, 32, 32, 0), (3129 * 2, 512, 512, 3129)])
def test_planes_to_planes_linear_is_the_encoder_applied_to_the_f32_product(rows, M, K, batch):
    """cti_gemm_nt_f16f6_planes without scale / bias / activation: its planes are, bit for bit, the encoder's output on the fp32 result of the
    same product taken in the same orientation (W X^T written transposed) -- the register epilogue's lane-pair exchange, block / row
    addressing and batch padding under one exact check."""
    g = torch.Generator().manual_seed(rows + M + K)
    x = torch.relu(torch.randn(rows, K, generator=g)) * 2.0
    w = torch.randn(M, K, generator=g) * 0.3
    px, pw = ops.quantize_f16f6(x.to(DEV)), ops.quantize_f16f6(w.to(DEV))
    y = ops.linear_f16f6_planes(px, rows, pw, M, K, batch_rows_out=batch)
    # the same accumulators as fp32: C^T[m, n] stored at n * M + m
    C = torch.empty(rows, M, device=DEV)
    L = cti_amd.pkg._lib
    L.check(L.lib().cti_gemm_nt_f16f6(pw.data_ptr(), M, 0, px.data_ptr(), rows, 0, C.data_ptr(), 1, M, 0, 1, 1, M, rows, K, 0, 1, 0, 0, ops._stream()), "g")
    ref = ops.quantize_f16f6(C, batch)
    got, want = unpack(y, rows, M, batch), unpack(ref, rows, M, batch)
    for a, b, name in zip(got, want, ("H", "lo codes", "hi scales", "lo scales")):
        assert np.array_equal(a.view(np.uint16) if a.dtype == np.float16 else a, b.view(np.uint16) if b.dtype == np.float16 else b), name
    err = np.abs(C.cpu().numpy() - (x.double() @ w.double().T).numpy()).max() / float((x.double() @ w.double().T).abs().max())
    assert err < 5e-5


@pytest.mark.parametrize("rows,M,K,hr", [(700, 128, 96, 16), (40000, 768, 64, 48)])
def test_planes_to_planes_linear_scale_bias_relu(rows, M, K, hr):
    """(the second shape has three row tiles and more tiles than workgroups: a workgroup's row tile changes mid-stream and it reloads its bias registers)"""
    g = torch.Generator().manual_seed(11)
    x = torch.relu(torch.randn(rows, K, generator=g))
    w = torch.randn(M, K, generator=g) * 0.2
    scale = torch.rand(M // hr, generator=g) + 0.5
    bias = torch.randn(M, generator=g) * 0.5
    pw = ops.quantize_f16f6(w.to(DEV), row_scale=scale.to(DEV), scale_div=hr)          # the weight-norm scale rides in the weight's block
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(unpack(pw, M, K), unpack(ops.quantize_f16f6((w * scale.repeat_interleave(hr)[:, None]).to(DEV)), M, K)))
    y = ops.linear_f16f6_planes(ops.quantize_f16f6(x.to(DEV)), rows, pw, M, K, batch_rows_out=70, bias=bias.to(DEV), relu=True)
    H, CL, SH, SL = unpack(y, rows, M, 70)
    ref = torch.relu((x.double() @ w.double().T) * scale.double().repeat_interleave(hr)[None, :] + bias.double()[None, :]).numpy()
    err = np.abs(_decode(H, CL, SL) - ref).max() / np.abs(ref).max()
    print("planes -> planes linear: normalised max error %.3g" % err)
    assert err < 5e-5
    assert (H >= 0).all()                                            # ReLU


def test_planes_to_planes_linear_out_of_range_values_match_the_encoder():
    """Values beyond f16's range (outside the format's domain): the register encoder's rarely taken branch gives the same planes as the
    stand-alone encoder -- hi part saturated, the excess in the residual's codes."""
    rows, M, K = 64, 32, 32
    x = torch.full((rows, K), 300.0)
    x[::3] = 0.01
    w = torch.full((M, K), 40.0)
    w[1::2] = -40.0
    px, pw = ops.quantize_f16f6(x.to(DEV)), ops.quantize_f16f6(w.to(DEV))
    y = ops.linear_f16f6_planes(px, rows, pw, M, K)
    C = torch.empty(rows, M, device=DEV)
    L = cti_amd.pkg._lib
    L.check(L.lib().cti_gemm_nt_f16f6(pw.data_ptr(), M, 0, px.data_ptr(), rows, 0, C.data_ptr(), 1, M, 0, 1, 1, M, rows, K, 0, 1, 0, 0, ops._stream()), "g")
    assert float(C.abs().max()) > 65504
    for a_, b_ in zip(unpack(y, rows, M), unpack(ops.quantize_f16f6(C), rows, M)):
        assert np.array_equal(a_.view(np.uint16) if a_.dtype == np.float16 else a_, b_.view(np.uint16) if b_.dtype == np.float16 else b_)
    yr = ops.linear_f16f6_planes(px, rows, pw, M, K, relu=True)
    for a_, b_ in zip(unpack(yr, rows, M), unpack(ops.quantize_f16f6(torch.relu(C)), rows, M)):
        assert np.array_equal(a_.view(np.uint16) if a_.dtype == np.float16 else a_, b_.view(np.uint16) if b_.dtype == np.float16 else b_)


@pytest.mark.parametrize("B,V,Q,R", [(3, 36, 14, 32), (2, 9, 5, 4), (5, 33, 16, 2), (1, 40, 8, 6)])
def test_m_build_encodes_its_planes_itself(B, V, Q, R):
    """cti_paralind_mbuild_f16f6_fwd (modes 1 + 2 of src/Tensor.py:6-13 straight into the mode-3 product's f16f6 block: two ranks per scale
    block, lane-pair exchange, wave-private hold buffer) against the exact-fp32 M build: hi + decoded lo within the format's 2^-15 of the
    block maximum plus the three-product bf16 arithmetic of the build, scales consistent with the decoded block, batches padded to 8 rows."""
    hr, G = 16, 2
    g = torch.Generator().manual_seed(B * 100 + V + Q + R)
    Vr = torch.relu(torch.randn(B, V, R * hr, generator=g)).to(DEV)
    Qr = torch.relu(torch.randn(B, Q, R * hr, generator=g)).to(DEV)
    Teff = torch.randn(R, hr, hr, hr, G, generator=g).to(DEV)
    M = ops.paralind_mbuild(Vr, Qr, Teff).reshape(B * V * Q * G, R * hr).cpu().numpy().astype(np.float64)
    blk = ops.paralind_mbuild_f16f6(Vr, Qr, Teff)
    H, CL, SH, SL = unpack(blk, B * V * Q * G, R * hr, V * Q * G)
    dec = _decode(H, CL, SL)
    bmax = np.repeat(np.abs(M).reshape(M.shape[0], -1, 32).max(-1), 32, axis=1)
    err = np.max(np.abs(dec - M) / np.maximum(bmax, 1e-30))
    print("M build -> f16f6 planes B=%d V=%d Q=%d R=%d: max error / block max %.3g" % (B, V, Q, R, err))
    assert err < 6e-5
    # the stored scales are the encoder's for the values actually stored
    h, ch, cl, sh, sl = np_encode(H.astype(np.float32))
    assert np.array_equal(SH, sh)
