#!/usr/bin/env python3
"""CPU emulation of candidate split-product schemes for the mode-3 GEMM (out = M . Ar^T, K = 512) at the BASELINE configs[1] widths,
against float64 truth.  Schemes:
  bf16x3 : hi/lo bf16 planes, ah*bh + ah*bl + al*bh  (what the library does today)
  f16f6  : a16*b16 (f16 MFMA) + fp6(a16)*fp6(b_lo) + fp6(a_lo)*fp6(b16)   (two block-scaled e2m3 correction products, 32-wide K blocks)
  f16f8  : the same with e4m3 corrections (per-tensor scale)
Operands: M and Ar of sample 0 of the g3_tcnet_forward_c2 fixture inputs (numpy oracle, float64)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from oracle import cti_oracle as O


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def e2m3(x):
    """round-to-nearest onto the e2m3 grid (max 7.5, spacing .125 below 2, .25 in [2,4), .5 in [4,8)); saturating."""
    ax = np.abs(x)
    step = np.where(ax < 2, 0.125, np.where(ax < 4, 0.25, 0.5))
    y = np.minimum(np.round(ax / step) * step, 7.5)
    return np.sign(x) * y


def e4m3(x):
    ax = np.abs(x).astype(np.float64)
    e = np.floor(np.log2(np.maximum(ax, 1e-300)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    y = np.minimum(np.round(ax / step) * step, 448.0)
    return np.sign(x) * y


def block_scale_e2m3(x, blk=32, shared_exp=None):
    """x (rows, K): per (row, 32-block) power-of-two scale s.t. max|x|/scale <= 7.5; returns (q, scale) with x ~= q*scale."""
    r, K = x.shape
    xb = x.reshape(r, K // blk, blk)
    mx = np.abs(xb).max(-1, keepdims=True)
    if shared_exp is None:
        e = np.ceil(np.log2(np.maximum(mx, 1e-300) / 7.5))       # smallest power of two with mx / 2^e <= 7.5
    else:
        e = shared_exp
    sc = 2.0 ** e
    return (e2m3(xb / sc)).reshape(r, K), np.broadcast_to(sc, xb.shape).reshape(r, K), e


def main():
    fx, params, v, q, a, idx = gu.c2_case()
    dt = np.float64
    p = O._sub(params, "TriAtt.")
    R, hr, G = O._tc_dims(p)
    b = 0
    vt = O.fcnet(v[b:b+1], p, "v_tucker.", dtype=dt); qt = O.fcnet(q[b:b+1], p, "q_tucker.", dtype=dt); at = O.fcnet(a[b:b+1], p, "a_tucker.", dtype=dt)
    Vr = O._rank_proj(vt, p, "v", R, dt); Qr = O._rank_proj(qt, p, "q", R, dt); Ar = O._rank_proj(at, p, "a", R, dt)
    Te = O.teff_from_tg(p["T_g"], dt)
    X = np.einsum("rijkg,bvri->bvrjkg", Te, Vr, optimize=True)
    M = np.einsum("bvrjkg,bqrj->bvqgrk", X, Qr, optimize=True).reshape(-1, R * hr)
    Arm = Ar.reshape(-1, R * hr)
    M32 = M.astype(np.float32).astype(np.float64); A32 = Arm.astype(np.float32).astype(np.float64)
    truth = M32 @ A32.T
    nrm = np.abs(truth).max()
    print("M %s absmax %.3g rms %.3g | Ar %s absmax %.3g rms %.3g zeros %.1f%% | out absmax %.3g" % (M.shape, np.abs(M).max(), np.sqrt((M**2).mean()),
          Arm.shape, np.abs(Arm).max(), np.sqrt((Arm**2).mean()), 100 * (Arm == 0).mean(), nrm))
    # fp32 reference arithmetic
    e32 = np.abs((M32.astype(np.float32) @ A32.astype(np.float32).T).astype(np.float64) - truth).max() / nrm
    # bf16x3
    ah = bf16(M32).astype(np.float64); al = bf16(M32 - ah).astype(np.float64)
    bh = bf16(A32).astype(np.float64); bl = bf16(A32 - bh).astype(np.float64)
    o = ah @ bh.T + ah @ bl.T + al @ bh.T
    print("fp32 matmul      err %.3g" % e32)
    print("bf16x3           err %.3g" % (np.abs(o - truth).max() / nrm))
    print("bf16 plain       err %.3g" % (np.abs(ah @ bh.T - truth).max() / nrm))
    # f16 hi (per-tensor power-of-two scaling into the f16 range)
    def f16(x):
        s = 2.0 ** np.floor(np.log2(1024.0 / np.abs(x).max()))       # absmax -> [512, 1024)
        return (x * s).astype(np.float16).astype(np.float64) / s
    a16 = f16(M32); b16 = f16(A32)
    al_ = M32 - a16; bl_ = A32 - b16
    print("f16 plain        err %.3g" % (np.abs(a16 @ b16.T - truth).max() / nrm))
    print("f16 + exact corr err %.3g" % (np.abs(a16 @ b16.T + a16 @ bl_.T + al_ @ b16.T - truth).max() / nrm))
    # f16f6: block-scaled e2m3 of hi and of lo (lo shares the hi block exponent - 11 or has its own)
    for own in (True, False):
        qa_h, sa_h, ea = block_scale_e2m3(a16); qb_h, sb_h, eb = block_scale_e2m3(b16)
        if own:
            qa_l, sa_l, _ = block_scale_e2m3(al_); qb_l, sb_l, _ = block_scale_e2m3(bl_)
        else:
            qa_l, sa_l, _ = block_scale_e2m3(al_, shared_exp=ea - 11); qb_l, sb_l, _ = block_scale_e2m3(bl_, shared_exp=eb - 11)
        o = a16 @ b16.T + (qa_h * sa_h) @ (qb_l * sb_l).T + (qa_l * sa_l) @ (qb_h * sb_h).T
        print("f16f6 (lo scale %s) err %.3g" % ("own" if own else "hi-11", np.abs(o - truth).max() / nrm))
    # f16f8: e4m3 corrections with per-tensor scales
    def q8(x):
        s = 2.0 ** np.floor(np.log2(256.0 / np.abs(x).max()))
        return e4m3(x * s) / s
    o = a16 @ b16.T + q8(a16) @ q8(bl_).T + q8(al_) @ q8(b16).T
    print("f16f8            err %.3g" % (np.abs(o - truth).max() / nrm))
    # one-sided variants
    o = a16 @ b16.T + a16 @ bl_.T
    print("f16 + (b lo only, exact) err %.3g" % (np.abs(o - truth).max() / nrm))


if __name__ == "__main__":
    main()
