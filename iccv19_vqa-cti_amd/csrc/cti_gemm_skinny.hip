// cti_gemm_skinny.hip -- the model forwards' SKINNY products (M = the batch's 256 rows: classifier layers, the residual / K-concatenated projections of the glimpse
// loops; reference src/classifier.py, src/fc.py:22-29 behind src/FFOE/base_model.py:53-67,121-136) as ONE launch each (round 6).
//
// Until round 5 such a product was: split the fp32 activations into planes (or read them through the LDS-DMA ring as fp32), S = 8-16 K ranges as extra workgroups of
// the 128 x 128 planes kernel writing fp32 partials [S][M][N], and a reduce / epilogue kernel -- 2-3 dependent launches of 5-16 us for 0.3-1 GFLOP, three of them in
// a row at the end of every forward.  Here a workgroup owns a 32 x 64 (or 32 x 32) output tile and its EIGHT waves an eighth of K each, for which a wave computes the
// whole tile: A fragments straight from the fp32 rows (two 16-B loads per lane and 32-deep step, converted -- or split into hi + lo -- in registers) or from operand
// planes, B fragments straight from the weight planes, all of a batch of steps in flight at once, no LDS staging, no barrier in the K loop; the eight partial tiles
// meet in LDS once, in wave order (deterministic), where thread (row, four columns) applies scale / bias / ReLU and stores 16 B.  The same idea as the GRU's
// K-split step kernel (cti_gru.hip) and the BiAttention logits (cti_attention.hip).
#include "cti_common.h"

#ifndef CTI_SKINNY_LB
#define CTI_SKINNY_LB 2          // launch bound: minimum waves per SIMD.  (Measured: 4 -- two workgroups per compute unit, <= 128 registers, with CTI_SKINNY_SB = 2 --
                                 // c3 476 -> 470 us but c4 1 344 -> 1 406 us: the long-K products of the BAN loop lose more to the extra round trips than co-residency returns)
#endif

namespace cti {
namespace {

typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));

struct GemmSkP {
    const float* Af; int64_t ldaf;                     // fp32 A rows (AF32) ...
    const unsigned short* Ah; const unsigned short* Al; int64_t pitchA;      // ... or A planes (elements per K chunk)
    const unsigned short* Bh; const unsigned short* Bl; int64_t pitchB;
    float* C; const float* scale; const float* bias;
    int64_t rA1, rB1, ldc_m, sC1, scale_bs, bias_bs;
    int M, N, nk, scale_div, relu;
};

__device__ __forceinline__ void sk_cvt(const float4 a, const float4 b, sk_bf16x8& hi, sk_bf16x8& lo, bool want_lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = static_cast<__bf16>(x[e]);
        if (want_lo) lo[e] = static_cast<__bf16>(x[e] - static_cast<float>(hi[e]));
    }
    if (!want_lo) lo = hi;
}

// TERMS: 1 = one bf16 product per pair, 3 = hi*hi + hi*lo + lo*hi (fp32-grade).  AF32: A = fp32 rows (else planes).  NT: 16-column tiles per workgroup (2 or 4).
template <int TERMS, bool AF32, int NT>
__global__ __launch_bounds__(512, (CTI_SKINNY_LB)) void gemm_skinny_kernel(GemmSkP p) {
    __shared__ __attribute__((aligned(16))) float part[8 * 2 * NT * 64 * 4];      // [wave][tile = mt * NT + j][lane][4]: 16 B per lane and tile
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * 32, z = blockIdx.z;
    const int spw = (p.nk + 7) / 8, s0 = wid * spw, my = max(0, min(p.nk, s0 + spw) - s0);
    // A fragment of step s: row m0 + 16 mt + lr (clamped: rows beyond M are computed and dropped), k = 32 s + 8 lq .. + 7
    const float* af[2]; const unsigned short* ah[2]; const unsigned short* al[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int64_t row = (int64_t)z * p.rA1 + min(m0 + mt * 16 + lr, p.M - 1);
        if (AF32) af[mt] = p.Af + row * p.ldaf + lq * 8;
        else {
            ah[mt] = p.Ah + (int64_t)(lq >> 1) * p.pitchA + row * 16 + (lq & 1) * 8;
            al[mt] = p.Al + (int64_t)(lq >> 1) * p.pitchA + row * 16 + (lq & 1) * 8;
        }
    }
    const unsigned short* bh[NT]; const unsigned short* bl[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int64_t row = (int64_t)z * p.rB1 + min(n0 + j * 16 + lr, p.N - 1);
        bh[j] = p.Bh + (int64_t)(lq >> 1) * p.pitchB + row * 16 + (lq & 1) * 8;
        bl[j] = p.Bl + (int64_t)(lq >> 1) * p.pitchB + row * 16 + (lq & 1) * 8;
    }
    sk_f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[mt][j] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef CTI_SKINNY_SB
#define CTI_SKINNY_SB 4
#endif
    constexpr int SB = (TERMS == 3 || NT == 4) ? 2 : CTI_SKINNY_SB;  // K steps whose fragments are in flight together
    for (int sb = 0; sb < my; sb += SB) {
        float4 ra[SB][2][2]; sk_bf16x8 pa[SB][2], pal[SB][2], pb[SB][NT], pbl[SB][NT];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            if (sb + i < my) {
                const int s = s0 + sb + i;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (AF32) { ra[i][mt][0] = *reinterpret_cast<const float4*>(af[mt] + (int64_t)s * 32); ra[i][mt][1] = *reinterpret_cast<const float4*>(af[mt] + (int64_t)s * 32 + 4); }
                    else {
                        pa[i][mt] = *reinterpret_cast<const sk_bf16x8*>(ah[mt] + (int64_t)s * 2 * p.pitchA);
                        if (TERMS == 3) pal[i][mt] = *reinterpret_cast<const sk_bf16x8*>(al[mt] + (int64_t)s * 2 * p.pitchA);
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    pb[i][j] = *reinterpret_cast<const sk_bf16x8*>(bh[j] + (int64_t)s * 2 * p.pitchB);
                    if (TERMS == 3) pbl[i][j] = *reinterpret_cast<const sk_bf16x8*>(bl[j] + (int64_t)s * 2 * p.pitchB);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            if (sb + i < my) {
                sk_bf16x8 xh[2], xl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (AF32) sk_cvt(ra[i][mt][0], ra[i][mt][1], xh[mt], xl[mt], TERMS == 3);
                    else { xh[mt] = pa[i][mt]; xl[mt] = TERMS == 3 ? pal[i][mt] : pa[i][mt]; }
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        // (B fragment first: the accumulator's registers then run along n -- 16-B stores)
                        if (TERMS == 3) {
                            acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb[i][j], xl[mt], acc[mt][j], 0, 0, 0);
                            acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pbl[i][j], xh[mt], acc[mt][j], 0, 0, 0);
                        }
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb[i][j], xh[mt], acc[mt][j], 0, 0, 0);
                    }
            }
        }
    }
    // acc[mt][j][e] = partial C[m0 + 16 mt + (lane & 15)][n0 + 16 j + 4 (lane >> 4) + e]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<sk_f32x4*>(part + ((size_t)(wid * 2 * NT + mt * NT + j) * 64 + lane) * 4) = acc[mt][j];
    __syncthreads();
    // thread -> (row r of 32, four columns 4 c .. + 3 of 16 NT): 32 x 4 NT items
    for (int it = t; it < 32 * 4 * NT; it += 512) {
        const int r = it / (4 * NT), c = it - r * (4 * NT);
        const int mt = r >> 4, j = c >> 2, src_lane = (c & 3) * 16 + (r & 15);
        const float* pp = part + ((size_t)(mt * NT + j) * 64 + src_lane) * 4;
        sk_f32x4 sum = *reinterpret_cast<const sk_f32x4*>(pp);
#pragma unroll
        for (int w = 1; w < 8; ++w) sum += *reinterpret_cast<const sk_f32x4*>(pp + (size_t)w * 2 * NT * 64 * 4);
        const int m = m0 + r, n = n0 + 4 * c;
        if (m >= p.M || n >= p.N) continue;
        const float* sc = p.scale ? p.scale + (int64_t)z * p.scale_bs : nullptr;
        const float* bi = p.bias ? p.bias + (int64_t)z * p.bias_bs : nullptr;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ne = min(n + e, p.N - 1);
            v[e] = sum[e] * (sc ? sc[ne / p.scale_div] : 1.f) + (bi ? bi[ne] : 0.f);
            if (p.relu) v[e] = relu_nan(v[e]);
        }
        float* o = p.C + (int64_t)z * p.sC1 + (int64_t)m * p.ldc_m + n;
        if (n + 3 < p.N && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        else
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < p.N) o[e] = v[e];
    }
}

}  // namespace

// CTI_GEMM_SKINNY=0: off everywhere (A/B)
bool gemm_skinny_enabled() { static const bool v = [] { const char* e = getenv("CTI_GEMM_SKINNY"); return !(e && e[0] == '0'); }(); return v; }

// One fp32-row product with few rows and a long K: what gemm_nt_planes() would cut into K ranges + a reduce launch (or run as a handful of tiles)
bool gemm_skinny_eligible(const PlaneGemmArgs& a) {
    if (!gemm_skinny_enabled() || !(a.terms == 1 || a.terms == 3) || a.epi != 0 || a.Abf || a.f6out || a.partials_only || a.ldc_n != 1) return false;
    if ((a.nb2 > 1) || a.nb1 < 1 || a.nb1 > 64 || a.M < 1 || a.M > 512 || a.N < 16 || a.Kp % 32 != 0 || a.Kp < 256) return false;
    if (a.Af && ((a.ldaf & 3) || (reinterpret_cast<uintptr_t>(a.Af) & 15) || a.Kreal != a.Kp)) return false;     // (no K tail in the fp32 rows: every step reads 32 real values)
    return true;
}

int gemm_skinny(const PlaneGemmArgs& a, hipStream_t st) {
    GemmSkP p{};
    p.Af = a.Af; p.ldaf = a.ldaf; p.Ah = a.Ah; p.Al = a.Al; p.pitchA = a.rows_allocA * 16;
    p.Bh = a.Bh; p.Bl = a.Bl; p.pitchB = a.rows_allocB * 16;
    p.C = a.C; p.scale = a.scale; p.bias = a.bias;
    p.rA1 = a.rA1; p.rB1 = a.rB1; p.ldc_m = a.ldc_m; p.sC1 = a.sC1; p.scale_bs = a.scale_bs; p.bias_bs = a.bias_bs;
    p.M = a.M; p.N = a.N; p.nk = a.Kp / 32; p.scale_div = a.scale_div > 0 ? a.scale_div : 1; p.relu = a.relu;
    // 32 x 64 tiles when they still give the chip a workgroup per compute unit, else 32 x 32
    const long long rows = (a.M + 31) / 32;
    // (measured and dropped: the 32 x 64 tile for K >= 4 096 whatever the workgroup count -- a third fewer operand bytes through the L2s, half the workgroups:
    // c4 1 369 -> 1 394 us)
    const bool wide = rows * ((a.N + 63) / 64) * a.nb1 >= 200;
    const dim3 grid((unsigned)((a.N + (wide ? 63 : 31)) / (wide ? 64 : 32)), (unsigned)rows, (unsigned)a.nb1);
#define SK_GO(T, F) { if (wide) hipLaunchKernelGGL((gemm_skinny_kernel<T, F, 4>), grid, dim3(512), 0, st, p); else hipLaunchKernelGGL((gemm_skinny_kernel<T, F, 2>), grid, dim3(512), 0, st, p); }
    if (a.terms == 3) { if (a.Af) SK_GO(3, true) else SK_GO(3, false) }
    else              { if (a.Af) SK_GO(1, true) else SK_GO(1, false) }
#undef SK_GO
    return launch_status("gemm_skinny");
}

}  // namespace cti
