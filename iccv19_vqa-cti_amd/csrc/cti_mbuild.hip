// cti_mbuild.hip -- modes 1 and 2 of the PARALIND core, fast path.
//
//   M[b, (v,q,g), r*HR + k] = sum_j Qr[b,q,r*HR+j] * ( sum_i T_eff[r,i,j,k,g] * Vr[b,v,r*HR+i] )
//
// One 1024-thread workgroup per (sample b, group of RG = 32/HR ranks).  Per rank: T_eff[r] (HR^3*G floats, 32 KiB at
// the default HR=16, G=2) and the sample's Vr/Qr slices are staged in LDS; step 1 builds X[v][g][j][k] in LDS
// (columns of T_eff read conflict-free, Vr broadcast as b128); step 2 gives every thread one output row (v,q,g)
// and 16 j x HR FMAs on b128 reads of X (rows ordered (v,g,q) across lanes so a 16-lane group shares its X row =
// LDS broadcast).  After RG ranks a thread holds 32 consecutive K-columns of its row and stores them as 8 (planes:
// 2 x 4) 16-B pieces, i.e. 64/128 contiguous bytes: M is written exactly once, in the layout the mode-3 GEMM DMA reads.
// 35 MFLOP/sample at C2, fp32 VALU (exact), HBM traffic = the M planes (2 MB/sample) + Vr/Qr (0.1 MB/sample).
#include "cti_common.h"
#include "cti_f16f6.h"

namespace cti {
namespace {

__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x)); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

constexpr int MB_ROWS = 2;       // step-2 work items (row pairs) per thread: V*G*ceil(Q/2) <= 2048

#ifndef CTI_MBF_SKIP
#define CTI_MBF_SKIP 0        // timing-only ablation mask (tools/tune_mbuild.py): 1 step 1, 2 step 2 arithmetic, 4 stores
#endif
template <int HR, bool PLANES>
__global__ __launch_bounds__(1024) void mbuild_fast_kernel(const float* __restrict__ Vr, const float* __restrict__ Qr,
                                                           const float* __restrict__ Teff, float* __restrict__ Mf,
                                                           unsigned short* __restrict__ Mh, unsigned short* __restrict__ Ml,
                                                           int V, int Q, int R, int G,
                                                           int64_t ldm /* fp32: row stride of M; planes: chunk pitch (elements) */,
                                                           int rpb /* ranks per workgroup */) {
    constexpr int HH = HR * HR;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int inner = HH * G;                    // columns (j,k,g) of T_eff[r][i]
    float* Ts = sm;                              // [HR][inner]
    float* Xs = Ts + HR * inner;                 // [V][G][HR(j)][HR(k)]
    float* Vs = Xs + (size_t)V * G * HH;         // [V][HR]
    float* Qs = Vs + (V + 8) * HR;               // [HR(j)][Qpad]  (8 slack rows behind Vs: step 1 reads v0..v0+7 unguarded)
    const int Qpad = (Q + 1) | 1;
    const int t = threadIdx.x;
    constexpr int nthr = 1024;
    const int b = blockIdx.y;
    const int K = R * HR;
    const int rows = V * Q * G;
    const float* vb = Vr + (int64_t)b * V * K;
    const float* qb = Qr + (int64_t)b * Q * K;
    const int r_lo = blockIdx.x * rpb, r_hi = min(R, r_lo + rpb);

    // ---- everything that does not depend on the rank is decoded ONCE (runtime integer divisions cost ~40 instructions
    // each; inside the rank loop they outweighed the FMAs 7:1) -----------------------------------------------------------
    // step-1 item: (column c of T_eff, v range [v0, v0 + vspan)); the 1024 threads cover inner columns x nsplit v ranges
    // (the v range of a thread is read through the scalar path, so it must be wave-uniform: with inner % 64 != 0 -- hr = 4 and
    // G not a multiple of 4, test-sized models only -- a wave would straddle two v ranges, so those shapes do not split v)
    const int nsplit = (inner % 64 == 0) ? max(1, nthr / inner) : 1, vspan = (V + nsplit - 1) / nsplit;
    int it_c[1], it_v0[1], it_x[1];
    {
        const int c = t % inner, sp = t / inner;
        const int g = c % G, k = (c / G) % HR, j = c / (G * HR);
        it_c[0] = (sp < nsplit && sp * vspan < V) ? c : -1;
        it_v0[0] = sp * vspan;
        it_x[0] = (sp * vspan * G + g) * HH + j * HR + k;       // X offset of (v0, g, j, k); +G*HH per v
    }
    // step-2 work item: rows (v, q, g) and (v, q+1, g); lane order (v, g, q-pair), q-pair fastest: a 16-lane group mostly
    // shares its X row (LDS broadcast)
    const int QP = (Q + 1) / 2, items2 = V * G * QP;
    int row_x[MB_ROWS], row_q[MB_ROWS];
    int64_t row_o[MB_ROWS];
#pragma unroll
    for (int n = 0; n < MB_ROWS; ++n) {
        const int lr = t + n * nthr;
        const int lq = (lr % QP) * 2, lg = (lr / QP) % G, lv = lr / (QP * G);
        row_q[n] = lr < items2 ? lq : -1;
        row_x[n] = (lv * G + lg) * HH;
        row_o[n] = (int64_t)b * rows + ((int64_t)lv * Q + lq) * G + lg;     // output row in the (v,q,g) order of the mode-3 GEMM
    }
    // operand loads of one rank: T_eff[r] (HR*inner floats, float4 per thread and trip), Vr/Qr slices (<= 1 element each)
    const int t4 = HR * inner / 4;                                             // float4 count of T_eff[r]
    const int ve = t < V * HR ? (t / HR) * K + (t % HR) : -1;                  // + r*HR
    const int qe = t < Q * HR ? (t / HR) * K + (t % HR) : -1;
    const int qdst = (t % HR) * Qpad + (t / HR);
    constexpr int T4MAX = 4;                                                    // HR*inner/4 <= 4096 float4 (HR=16, G<=4)
    static_assert(T4MAX == 4, "the prefetch registers below are written out by hand");
    float4 tp0 = make_float4(0.f, 0.f, 0.f, 0.f), tp1 = tp0, tp2 = tp0, tp3 = tp0;     // named registers, not an array
    float vpre = 0.f, qpre = 0.f;
#define CTI_MB_PREFETCH(rr)                                                                                   \
    {                                                                                                         \
        const float4* Tr_ = reinterpret_cast<const float4*>(Teff + (int64_t)(rr) * HR * inner);               \
        if (t < t4) tp0 = Tr_[t];                                                                             \
        if (t + nthr < t4) tp1 = Tr_[t + nthr];                                                               \
        if (t + 2 * nthr < t4) tp2 = Tr_[t + 2 * nthr];                                                       \
        if (t + 3 * nthr < t4) tp3 = Tr_[t + 3 * nthr];                                                       \
        if (ve >= 0) vpre = vb[ve + (rr) * HR];                                                               \
        if (qe >= 0) qpre = qb[qe + (rr) * HR];                                                               \
    }
    CTI_MB_PREFETCH(r_lo)

    for (int r = r_lo; r < r_hi; ++r) {
        __syncthreads();                                      // previous rank's readers are done with Ts/Xs/Vs/Qs
        if (t < t4) reinterpret_cast<float4*>(Ts)[t] = tp0;
        if (t + nthr < t4) reinterpret_cast<float4*>(Ts)[t + nthr] = tp1;
        if (t + 2 * nthr < t4) reinterpret_cast<float4*>(Ts)[t + 2 * nthr] = tp2;
        if (t + 3 * nthr < t4) reinterpret_cast<float4*>(Ts)[t + 3 * nthr] = tp3;
        if (ve >= 0) Vs[t] = vpre;
        if (qe >= 0) Qs[qdst] = qpre;
        __syncthreads();
        if (r + 1 < r_hi) CTI_MB_PREFETCH(r + 1)              // next rank's operands fly under this rank's FMAs
        // step 1:  X[v][g][j][k] = sum_i T_eff[r][i][j,k,g] * Vr[v][i].  A thread owns ONE column c = (j,k,g) of T_eff[r] (its HR
        // values stay in registers) and half of the v range; Vr[v][0..HR) is wave-uniform, so it is read from global memory
        // through the scalar path (s_load -> SGPR operands of the FMAs): no LDS broadcast traffic at all.
        {
            const int c = it_c[0];
            if (c >= 0 && !(CTI_MBF_SKIP & 1)) {
                float tc[HR];
#pragma unroll
                for (int i = 0; i < HR; ++i) tc[i] = Ts[i * inner + c];
                float* xo = Xs + it_x[0];
                const int vlo = __builtin_amdgcn_readfirstlane(it_v0[0]), vhi = min(V, vlo + vspan);   // wave-uniform (inner % 64 == 0)
                const float* vsrc = vb + r * HR;
                for (int v = vlo; v < vhi; ++v) {
                    const float* vr = vsrc + (int64_t)v * K;                  // uniform address
                    float x = 0.f;
#pragma unroll
                    for (int i = 0; i < HR; ++i) x = fmaf(tc[i], vr[i], x);
                    xo[(v - vlo) * G * HH] = x;
                }
            }
        }
        __syncthreads();
        // step 2: output rows (v,q,g) and (v,q+1,g) per thread share the X row: HR columns each
#pragma unroll
        for (int n = 0; n < MB_ROWS; ++n) {
            const int lq = row_q[n];
            if (lq >= 0) {
            const bool two = lq + 1 < Q;
            float acc[HR], acc2[HR];
#pragma unroll
            for (int k = 0; k < HR; ++k) { acc[k] = 0.f; acc2[k] = 0.f; }
            const float* xr = Xs + row_x[n];
#pragma unroll 2
            for (int j = 0; j < ((CTI_MBF_SKIP & 2) ? 0 : HR); ++j) {
                const float qv = Qs[j * Qpad + lq];
                const float qw = Qs[j * Qpad + lq + 1];                     // Qpad >= Q + 1 columns: in bounds, unused when !two
#pragma unroll
                for (int k4 = 0; k4 < HR; k4 += 4) {
                    const float4 xx = *reinterpret_cast<const float4*>(xr + j * HR + k4);
                    acc[k4 + 0] = fmaf(qv, xx.x, acc[k4 + 0]); acc2[k4 + 0] = fmaf(qw, xx.x, acc2[k4 + 0]);
                    acc[k4 + 1] = fmaf(qv, xx.y, acc[k4 + 1]); acc2[k4 + 1] = fmaf(qw, xx.y, acc2[k4 + 1]);
                    acc[k4 + 2] = fmaf(qv, xx.z, acc[k4 + 2]); acc2[k4 + 2] = fmaf(qw, xx.z, acc2[k4 + 2]);
                    acc[k4 + 3] = fmaf(qv, xx.w, acc[k4 + 3]); acc2[k4 + 3] = fmaf(qw, xx.w, acc2[k4 + 3]);
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
            if (half == 1 && !two) break;
            if ((CTI_MBF_SKIP & 4) && acc[0] != 12345.f) break;
            const float* av = half ? acc2 : acc;
            const int64_t orow = row_o[n] + (half ? G : 0);                 // (v, q+1, g) is G rows further
            const int c0 = r * HR;
            if (PLANES) {
                // chunk-major planes: column c of row orow lives at (c >> 4) * pitch + orow * 16 + (c & 15); the rows of one
                // v are 32 B apart, so a wave's stores fall in a few contiguous KiB
                const int64_t o = (int64_t)(c0 >> 4) * ldm + orow * 16 + (c0 & 15);
                unsigned short* ph = Mh + o;
                unsigned short* pl = Ml + o;
                auto pk = [](float x0, float x1, unsigned& hi2, unsigned& lo2) {
                    const unsigned short h0 = bf16_bits(x0), h1 = bf16_bits(x1);
                    hi2 = h0 | ((unsigned)h1 << 16);
                    lo2 = bf16_bits(x0 - bf16_to_f32(h0)) | ((unsigned)bf16_bits(x1 - bf16_to_f32(h1)) << 16);
                };
                if (HR % 8 == 0) {
#pragma unroll
                    for (int c8 = 0; c8 < HR / 8; ++c8) {
                        uint4 hq, lq4;
                        pk(av[c8 * 8 + 0], av[c8 * 8 + 1], hq.x, lq4.x); pk(av[c8 * 8 + 2], av[c8 * 8 + 3], hq.y, lq4.y);
                        pk(av[c8 * 8 + 4], av[c8 * 8 + 5], hq.z, lq4.z); pk(av[c8 * 8 + 6], av[c8 * 8 + 7], hq.w, lq4.w);
                        *reinterpret_cast<uint4*>(ph + c8 * 8) = hq;
                        *reinterpret_cast<uint4*>(pl + c8 * 8) = lq4;
                    }
                } else {                                    // HR == 4: 8-byte pieces
                    uint2 hq, lq2;
                    pk(av[0], av[1], hq.x, lq2.x); pk(av[2], av[3], hq.y, lq2.y);
                    *reinterpret_cast<uint2*>(ph) = hq;
                    *reinterpret_cast<uint2*>(pl) = lq2;
                }
                if (r == R - 1) {                           // zero the K tail [K, Kp) of the planes (Kp = K rounded up to 32)
                    for (int c = K; c < ((K + 31) & ~31); ++c) {
                        const int64_t oz = (int64_t)(c >> 4) * ldm + orow * 16 + (c & 15);
                        Mh[oz] = 0; Ml[oz] = 0;
                    }
                }
            } else {
                float* pf = Mf + orow * ldm + c0;
                if ((ldm & 3) == 0) {
#pragma unroll
                    for (int k4 = 0; k4 < HR; k4 += 4) *reinterpret_cast<float4*>(pf + k4) = make_float4(av[k4], av[k4 + 1], av[k4 + 2], av[k4 + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < HR; ++k) pf[k] = av[k];
                }
            }
            }                                                             // half
            }
        }
    }
}

template <int HR, bool PLANES>
int launch(const float* Vr, const float* Qr, const float* Teff, float* Mf, unsigned short* Mh, unsigned short* Ml, int B, int V,
           int Q, int R, int G, int64_t ldm, size_t lds, hipStream_t st) {
    auto kern = mbuild_fast_kernel<HR, PLANES>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail((int)e, "mbuild_fast: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    // one workgroup per CU at a time (LDS); split the ranks over enough workgroups to cover the 256 CUs
    int groups = (256 + B - 1) / B;
    if (groups > R) groups = R;
    const int rpb = (R + groups - 1) / groups;
    dim3 grid((R + rpb - 1) / rpb, B);
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, Vr, Qr, Teff, Mf, Mh, Ml, V, Q, R, G, ldm, rpb);
    return launch_status("mbuild_fast");
}

}  // namespace

// returns CTI_E_UNSUPPORTED (without setting an error message the caller must surface) when the shape is outside the
// fast path, so that the caller can take the generic kernel of cti_paralind.hip.
static size_t mbuild_fast_lds(int V, int Q, int hr, int G) {
    return sizeof(float) * ((size_t)hr * hr * hr * G + (size_t)V * G * hr * hr + (size_t)(V + 8) * hr + (size_t)hr * ((Q + 1) | 1));
}
// the fast kernel's shape test by sizes only (cti_tcnet.hip plans its workspace with it)
bool mbuild_fast_fits(int B, int V, int Q, int R, int hr, int G) {
    (void)R;
    if (hr != 4 && hr != 8 && hr != 16) return false;
    if (B > 65535 || mbuild_fast_lds(V, Q, hr, G) > 160 * 1024) return false;
    // per-thread column / row / prefetch budgets of the fast kernel
    return !(hr * hr * G > 1024 || (int64_t)V * G * ((Q + 1) / 2) > 2048 || hr * hr * hr * G / 4 > 4096 || V * hr > 1024 || Q * hr > 1024);
}
int mbuild_fast(const float* Vr, const float* Qr, const float* Teff, float* Mf, unsigned short* Mh, unsigned short* Ml, int B,
                int V, int Q, int R, int hr, int G, int64_t ldm, hipStream_t st) {
    if (!mbuild_fast_fits(B, V, Q, R, hr, G)) return CTI_E_UNSUPPORTED;
    const size_t lds = mbuild_fast_lds(V, Q, hr, G);
    const bool planes = Mh != nullptr;
#define CTI_MB(H) (planes ? launch<H, true>(Vr, Qr, Teff, Mf, Mh, Ml, B, V, Q, R, G, ldm, lds, st) \
                          : launch<H, false>(Vr, Qr, Teff, Mf, Mh, Ml, B, V, Q, R, G, ldm, lds, st))
    switch (hr) {
        case 4: return CTI_MB(4);
        case 8: return CTI_MB(8);
        default: return CTI_MB(16);
    }
#undef CTI_MB
}

}  // namespace cti

// =====================================================================================================================
// M build on the MFMA (fp32-grade 3-product bf16 mode; hr = 16, G = 2, V <= 48, Q <= 16, even R: every model configuration).
// Both contractions of a rank are K = 16 GEMMs -- ONE 32x32x16 MFMA step each:
//   step 1  X[v, c=(j,k,g)] = sum_i Vr[v,i] T[i,c]        rows v (two 32-row tiles), 16 column tiles (j = tile index)
//   step 2  Mt[(g,k), q]    = sum_j X[v,(j,k,g)] Qr[q,j]  one tile per v: rows (g,k) = 32, columns q
// Fragments are "8 consecutive k of one row", and Vr / Qr / the PRE-TRANSPOSED core Tt[r][c][i] have the contraction axis
// contiguous in global memory, so their fragments are two 16-B global loads per lane (issued one rank ahead); only X changes
// hands through LDS ([v][g][k][j], pitch 20 floats: conflict-free b128 fragment reads).  Step 2 is oriented so that a lane's
// accumulator holds 4 CONSECUTIVE k of one output row (v,q,g): the hi/lo plane stores are 8 B each, no output staging.
// The VALU kernel above spends ~29k cycles per rank and workgroup; this one is bounded by the hi/lo splits (~5k).
// =====================================================================================================================
namespace cti {
namespace {

typedef __bf16 mb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float mb_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mb_split8(const float4 a, const float4 b, mb_bf16x8& hi, mb_bf16x8& lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = static_cast<__bf16>(x[e]);
        hi[e] = h;
        lo[e] = static_cast<__bf16>(x[e] - static_cast<float>(h));
    }
}

#ifndef CTI_MM_SKIP
#define CTI_MM_SKIP 0     // timing-only ablation mask: 1 plane stores, 2 X2 writes, 4 step-2 MFMAs + splits, 8 step-1 MFMAs
#endif
__device__ __forceinline__ uint4 mb_pack8(const unsigned short* h) {
    uint4 r;
    r.x = h[0] | ((unsigned)h[1] << 16); r.y = h[2] | ((unsigned)h[3] << 16);
    r.z = h[4] | ((unsigned)h[5] << 16); r.w = h[6] | ((unsigned)h[7] << 16);
    return r;
}
constexpr int MB_SP = 36;                       // pitch of the wave-private output patch [q][32 rho + 4 pad]
constexpr int MB_XP = 20;                       // X row pitch in floats (16 j + 4 pad: 80-B rows, 16-B aligned, conflict-free)

// X (the step-1 accumulators of both row tiles) -> LDS, branch-free: one base per row tile, re-derived every rank behind an opaque barrier,
// compile-time offsets, rows beyond V redirected to a pad column of row 0.  (The straightforward form -- 32 guarded scalar stores -- makes hipcc
// hoist 32 per-lane addresses out of the rank loop and spill them, and costs 32 exec branches per rank.)
__device__ __forceinline__ void mb_store_x(float* X2, const mb_f32x16& x0, const mb_f32x16& x1, int V, int kg, int xg, int xk, int wid, int lane) {
    constexpr int HR = 16, G = 2;
    float* xw = X2 + ((4 * kg * G + xg) * HR + xk) * MB_XP + wid;
    asm volatile("" : "+v"(xw));
    float* const trash = X2 + 16 + (lane & 3);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int vo = (e & 3) + 8 * (e >> 2), vv = vo + 4 * kg;
        float* d0 = xw + vo * (G * HR * MB_XP);
        if (V < 32) d0 = vv < V ? d0 : trash;
        *d0 = x0[e];
    }
    if (V > 32) {
        float* xw1 = xw + 32 * (G * HR * MB_XP);
        asm volatile("" : "+v"(xw1));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int vo = (e & 3) + 8 * (e >> 2), vv = vo + 4 * kg + 32;
            float* d1 = xw1 + vo * (G * HR * MB_XP);
            d1 = vv < V ? d1 : trash;
            *d1 = x1[e];
        }
    }
}

// F32OUT: the rows are written as fp32 (row stride pitchM floats) instead of bf16 hi/lo planes -- the f16f6 mode encodes them in one pass
template <bool F32OUT>
__global__ __launch_bounds__(1024) void mbuild_mfma_kernel(const float* __restrict__ Vr, const float* __restrict__ Qr,
                                                           const float* __restrict__ Tt, unsigned short* __restrict__ Mh,
                                                           unsigned short* __restrict__ Ml, float* __restrict__ Mf, int V, int Q, int R, int64_t pitchM) {
    constexpr int HR = 16, G = 2, INNER = HR * HR * G;          // 512 columns c = (j*16 + k)*2 + g
    extern __shared__ __attribute__((aligned(16))) float X2[];  // [V][G][HR(k)][MB_XP], then 16 output patches [16][MB_SP]
    float* stage = X2 + (size_t)V * G * HR * MB_XP;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;  // 16 waves
    const int l31 = lane & 31, kg = lane >> 5;
    const int K = R * HR;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* vb = Vr + (int64_t)b * V * K + kg * 8;
    const float* qb = Qr + (int64_t)b * Q * K + kg * 8;
    const int v0 = l31, v1 = 32 + l31;                          // step-1 A rows of the two tiles
    const bool v0ok = v0 < V, v1ok = v1 < V, qok = l31 < Q;
    const int c1 = wid * 32 + l31;                              // step-1 column of this lane: j = wid, (k, g) = (l31 >> 1, l31 & 1)
    const int xk = l31 >> 1, xg = l31 & 1;
    const int64_t rows_b = (int64_t)b * V * Q * G;
    // fragments of rank r, loaded one rank ahead
    float4 a00 = z4, a01 = z4, a10 = z4, a11 = z4, t0 = z4, t1 = z4, q0 = z4, q1 = z4;
#define CTI_MM_LOAD(rr)                                                                                             \
    {                                                                                                               \
        const int o_ = (rr) * HR;                                                                                   \
        if (v0ok) { a00 = *reinterpret_cast<const float4*>(vb + (int64_t)v0 * K + o_); a01 = *reinterpret_cast<const float4*>(vb + (int64_t)v0 * K + o_ + 4); } \
        if (v1ok) { a10 = *reinterpret_cast<const float4*>(vb + (int64_t)v1 * K + o_); a11 = *reinterpret_cast<const float4*>(vb + (int64_t)v1 * K + o_ + 4); } \
        if (qok)  { q0 = *reinterpret_cast<const float4*>(qb + (int64_t)l31 * K + o_); q1 = *reinterpret_cast<const float4*>(qb + (int64_t)l31 * K + o_ + 4); } \
        const float* tp_ = Tt + ((int64_t)(rr) * INNER + c1) * HR + kg * 8;                                         \
        t0 = *reinterpret_cast<const float4*>(tp_); t1 = *reinterpret_cast<const float4*>(tp_ + 4);                 \
    }
    CTI_MM_LOAD(0)
    for (int r = 0; r < R; ++r) {
        // ---- step 1: this wave's column tile (j = wid) for both row tiles ------------------------------------------------
        mb_bf16x8 ah0, al0, ah1, al1, th, tl, qh, ql;
        mb_split8(a00, a01, ah0, al0);
        mb_split8(a10, a11, ah1, al1);
        mb_split8(t0, t1, th, tl);
        mb_split8(q0, q1, qh, ql);
        if (r + 1 < R) CTI_MM_LOAD(r + 1)
        mb_f32x16 x0, x1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { x0[e] = 0.f; x1[e] = 0.f; }
        if (!(CTI_MM_SKIP & 8)) {
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, th, x0, 0, 0, 0);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, tl, x0, 0, 0, 0);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, th, x0, 0, 0, 0);
        }
        if (V > 32 && !(CTI_MM_SKIP & 8)) {
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, th, x1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, tl, x1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, th, x1, 0, 0, 0);
        }
        __syncthreads();                                        // step-2 readers of the previous rank are done with X2
        if (CTI_MM_SKIP & 2) { if (x0[0] == 12345.f) X2[0] = x1[0]; }
        else mb_store_x(X2, x0, x1, V, kg, xg, xk, wid, lane);
        __syncthreads();
        // ---- step 2: one tile per v: rows rho = g*16 + k, columns q ------------------------------------------------------
        const int sg = l31 >> 4, sk = l31 & 15;
        for (int v = wid; v < ((CTI_MM_SKIP & 4) ? 0 : V); v += 16) {
            const float* xr = X2 + ((v * G + sg) * HR + sk) * MB_XP + kg * 8;
            mb_bf16x8 xh, xl;
            mb_split8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), xh, xl);
            mb_f32x16 m;
#pragma unroll
            for (int e = 0; e < 16; ++e) m[e] = 0.f;
            m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, qh, m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ql, m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, qh, m, 0, 0, 0);
            // The tile goes through a wave-private LDS patch [q][rho] so that the stores are whole 64-B lines: the hi (or lo) bytes
            // of output rows (v,q,0) and (v,q,1) of this rank's chunk are contiguous (2 x 16 bf16), i.e. rho = g*16 + k in order.
            // 4 lanes x 16 B per (v,q): one store instruction per plane and tile (8-B pieces cost 8x the L2 write transactions).
            float* stg = stage + wid * (16 * MB_SP);
            if (qok) {
#pragma unroll
                for (int eg = 0; eg < 4; ++eg)
                    *reinterpret_cast<float4*>(stg + l31 * MB_SP + 8 * eg + 4 * kg) = make_float4(m[eg * 4], m[eg * 4 + 1], m[eg * 4 + 2], m[eg * 4 + 3]);
            }
            // (same wave writes and reads: LDS operations of one wave complete in order)
            const int sq = lane >> 2, sp = lane & 3;
            if (sq < Q && !((CTI_MM_SKIP & 1) && m[0] != 12345.f)) {
                const float4 y0 = *reinterpret_cast<const float4*>(stg + sq * MB_SP + sp * 8);
                const float4 y1 = *reinterpret_cast<const float4*>(stg + sq * MB_SP + sp * 8 + 4);
                const int64_t row = rows_b + ((int64_t)v * Q + sq) * G;                       // row of g = 0; g = 1 is the next row
                if (F32OUT) {                                   // lane sp holds rho = 8 sp .. 8 sp + 7 = (g = sp >> 1, k = 8 (sp & 1) ..): 32 B of row (v, q, g)
                    float* dst = Mf + (row + (sp >> 1)) * pitchM + r * HR + (sp & 1) * 8;
                    *reinterpret_cast<float4*>(dst) = y0; *reinterpret_cast<float4*>(dst + 4) = y1;
                    continue;
                }
                const float ys[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
                unsigned short hb[8], lb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { hb[u] = bf16_bits(ys[u]); lb[u] = bf16_bits(ys[u] - bf16_to_f32(hb[u])); }
                const int64_t o = (int64_t)r * pitchM + row * 16 + sp * 8;
                *reinterpret_cast<uint4*>(Mh + o) = mb_pack8(hb);
                *reinterpret_cast<uint4*>(Ml + o) = mb_pack8(lb);
            }
        }
    }
#undef CTI_MM_LOAD
#undef CTI_MM_LANE
}

// =====================================================================================================
// The same M build writing the f16f6 operand planes of the mode-3 product DIRECTLY (cti_f16f6.h): no fp32 M in HBM (528 MB written and read
// back at BASELINE configs[1]) and no encoding pass.  A scale block is 32 consecutive K = TWO ranks of one output row (v, q, g), and a
// step-2 tile holds one rank's 16 k for both g, spread over a lane pair exactly like the transposed GEMM's accumulators: eight
// v_permlane32_swap hand the lower lanes the 16 values of (v, q, g = 0) and the upper lanes those of g = 1.  The even rank's values wait
// in a wave-private LDS buffer (hold[v][4 chunks][(g, q)][4 floats]: a lane's four 16-B pieces are 16 B apart across lanes -- conflict-free;
// LDS has no room for a second rank of X), the odd rank's stay in registers, and every lane with a real q encodes one whole (row, block)
// item in registers (f6_encode_row32_regs; without the saturation-excess branch -- M values beyond f16's range, outside the format's
// domain, clamp to +-65504 here).  LDS: X2 + hold = (2560 + 128 Q) V bytes <= 160 KiB (configs[1]: 153 KiB).
#ifndef CTI_MBF6_PREFETCH_EARLY
#define CTI_MBF6_PREFETCH_EARLY 1
#endif
#ifndef CTI_MBF6_ABL      // timing-only ablations of mbuild_mfma_f6_kernel (wrong results; tools/tune_mbuild_f6.py): 1 the step-1 operand splits (V^, Q^, T) become four
#define CTI_MBF6_ABL 0    // byte permutes each, 2 the same for the step-2 X fragment, 4 no encoder (the odd rank's items are not encoded or stored), 8 no MFMAs
#endif
// (ablation helper) 8 floats -> the upper halves of their bit patterns as "hi" (truncation: 4 v_perm_b32), lo = hi: what a split would cost if it were free
__device__ __forceinline__ void mb_trunc8(const float4 a, const float4 b, mb_bf16x8& hi, mb_bf16x8& lo) {
    const unsigned x[8] = {__builtin_bit_cast(unsigned, a.x), __builtin_bit_cast(unsigned, a.y), __builtin_bit_cast(unsigned, a.z), __builtin_bit_cast(unsigned, a.w),
                           __builtin_bit_cast(unsigned, b.x), __builtin_bit_cast(unsigned, b.y), __builtin_bit_cast(unsigned, b.z), __builtin_bit_cast(unsigned, b.w)};
    typedef unsigned mb_u32x4 __attribute__((ext_vector_type(4)));
    mb_u32x4 p;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = (x[2 * i] >> 16) | (x[2 * i + 1] & 0xffff0000u);
    hi = __builtin_bit_cast(mb_bf16x8, p);
    lo = hi;
}
__global__ __launch_bounds__(1024) void mbuild_mfma_f6_kernel(const float* __restrict__ Vr, const float* __restrict__ Qr, const float* __restrict__ Tt,
                                                              F6Planes P, int V, int Q, int R) {
    constexpr int HR = 16, G = 2, INNER = HR * HR * G;
    extern __shared__ __attribute__((aligned(16))) float X2[];
    float* hold = X2 + (size_t)V * G * HR * MB_XP;
    const int b = blockIdx.x;
    const int lane0 = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: its multiples live in SGPRs, not in (spilled) VGPRs
    const int K = R * HR;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // uniform bases + 32-bit per-lane element offsets (global loads with an SGPR base): four 64-bit per-lane pointers less in a kernel at its register ceiling
    const float* vb = Vr + (int64_t)b * V * K;
    const float* qb = Qr + (int64_t)b * Q * K;
    const int64_t rows_b = (int64_t)b * V * Q * G;
    // Lane coordinates are re-derived from an opaque copy wherever a rank iteration needs them (round 3): as loop invariants the per-lane offsets of both steps
    // were hoisted out of the rank loop, 27 registers of them spilled in a kernel at its 128-register ceiling, and a scratch reload waits for vmcnt(0).
#define CTI_MM_LANE()                                                                                                    \
    int lane = lane0;                                                                                                    \
    asm volatile("" : "+v"(lane));                                                                                       \
    const int l31 = lane & 31, kg = lane >> 5;                                                                           \
    const unsigned ov0 = (unsigned)(l31 * K + kg * 8), ov1 = (unsigned)((32 + l31) * K + kg * 8), ot = (unsigned)((wid * 32 + l31) * HR + kg * 8); \
    const bool v0ok = l31 < V, v1ok = 32 + l31 < V, qok = l31 < Q;
    typedef float mbf_f32x4 __attribute__((ext_vector_type(4)));
    float4 a00 = z4, a01 = z4, a10 = z4, a11 = z4, t0 = z4, t1 = z4, q0 = z4, q1 = z4;
#define CTI_MM_LOAD(rr)                                                                                             \
    {                                                                                                               \
        CTI_MM_LANE()                                                                                               \
        const unsigned o_ = (unsigned)((rr) * HR);                                                                  \
        if (v0ok) { a00 = *reinterpret_cast<const float4*>(vb + (ov0 + o_)); a01 = *reinterpret_cast<const float4*>(vb + (ov0 + o_ + 4)); } \
        if (v1ok) { a10 = *reinterpret_cast<const float4*>(vb + (ov1 + o_)); a11 = *reinterpret_cast<const float4*>(vb + (ov1 + o_ + 4)); } \
        if (qok)  { q0 = *reinterpret_cast<const float4*>(qb + (ov0 + o_)); q1 = *reinterpret_cast<const float4*>(qb + (ov0 + o_ + 4)); } \
        const float* tp_ = Tt + (size_t)(rr) * INNER * HR;                                                          \
        t0 = *reinterpret_cast<const float4*>(tp_ + ot); t1 = *reinterpret_cast<const float4*>(tp_ + (ot + 4));     \
    }
    CTI_MM_LOAD(0)
    for (int r = 0; r < R; ++r) {
        mb_bf16x8 ah0, al0, ah1, al1, th, tl, qh, ql;
        if (CTI_MBF6_ABL & 1) { mb_trunc8(a00, a01, ah0, al0); mb_trunc8(a10, a11, ah1, al1); mb_trunc8(t0, t1, th, tl); mb_trunc8(q0, q1, qh, ql); }
        else {
        mb_split8(a00, a01, ah0, al0);
        mb_split8(a10, a11, ah1, al1);
        mb_split8(t0, t1, th, tl);
        mb_split8(q0, q1, qh, ql);
        }
#if CTI_MBF6_PREFETCH_EARLY
        if (r + 1 < R) CTI_MM_LOAD(r + 1)
#endif
        mb_f32x16 x0, x1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { x0[e] = 0.f; x1[e] = 0.f; }
        if (!(CTI_MBF6_ABL & 8) || R < 0) {
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, th, x0, 0, 0, 0);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, tl, x0, 0, 0, 0);
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, th, x0, 0, 0, 0);
        } else { x0[0] = static_cast<float>(ah0[0]) + static_cast<float>(th[1]) + static_cast<float>(al0[2]) + static_cast<float>(tl[3]); x1[0] = static_cast<float>(ah1[0]) + static_cast<float>(al1[1]); }
        if (V > 32 && (!(CTI_MBF6_ABL & 8) || R < 0)) {
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, th, x1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, tl, x1, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, th, x1, 0, 0, 0);
        }
        __syncthreads();
        CTI_MM_LANE()
        (void)ov0; (void)ov1; (void)ot; (void)v0ok; (void)v1ok;
        const int xk = l31 >> 1, xg = l31 & 1;
        mb_store_x(X2, x0, x1, V, kg, xg, xk, wid, lane);
        __syncthreads();
        const int sg = l31 >> 4, sk = l31 & 15;
        const bool odd = r & 1;
        for (int v = wid; v < V; v += 16) {
            const float* xr = X2 + ((v * G + sg) * HR + sk) * MB_XP + kg * 8;
            mb_bf16x8 xh, xl;
            if (CTI_MBF6_ABL & 2) mb_trunc8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), xh, xl);
            else mb_split8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), xh, xl);
            mb_f32x16 m;
#pragma unroll
            for (int e = 0; e < 16; ++e) m[e] = 0.f;
            if (!(CTI_MBF6_ABL & 8) || R < 0) {
            m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, qh, m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ql, m, 0, 0, 0);
            m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, qh, m, 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { m[e] = static_cast<float>(xh[e]) * static_cast<float>(qh[e]); m[8 + e] = static_cast<float>(xl[e]) + static_cast<float>(ql[e]); }
            }
            // register 4 eg + t of lane (q, kg) is rho = 8 eg + 4 kg + t = (g = eg >> 1, k = 8 (eg & 1) + 4 kg + t): the swaps of (eg 0, eg 2) and
            // (eg 1, eg 3) leave lane half g with y[k], k = 0 .. 15, of output row (v, q, g)
            float y[16];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float xa = m[t], ya = m[8 + t], xb = m[4 + t], yb = m[12 + t];
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(xa), "+v"(ya));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(xb), "+v"(yb));
                y[t] = xa; y[4 + t] = ya; y[8 + t] = xb; y[12 + t] = yb;
            }
            if (!qok) continue;
            float* hp = hold + ((size_t)v * 4 * (2 * Q) + kg * Q + l31) * 4;           // chunk c at + c * 2Q * 4 floats
            if (!odd) {
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<mbf_f32x4*>(hp + c * (2 * Q) * 4) = mbf_f32x4{y[4 * c], y[4 * c + 1], y[4 * c + 2], y[4 * c + 3]};
                continue;
            }
            float x[32];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const mbf_f32x4 hv = *reinterpret_cast<const mbf_f32x4*>(hp + c * (2 * Q) * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) x[4 * c + u] = hv[u];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) x[16 + k] = y[k];
            if ((CTI_MBF6_ABL & 4) && x[0] != 12345.f) continue;
            const int64_t prow = f6_prow(P, rows_b + ((int64_t)v * Q + l31) * G + kg);
            const int kb = r >> 1;
            const int64_t o = (int64_t)kb * P.rows_alloc + prow;
            f6_encode_row32_regs<false>(x, -__builtin_huge_valf(), reinterpret_cast<char*>(P.H) + o * 64, reinterpret_cast<char*>(P.FL) + o * 24,
                                 reinterpret_cast<char*>(P.S) + ((int64_t)kb * P.rows_allocS + prow) * 2);
        }
#if !CTI_MBF6_PREFETCH_EARLY
        if (r + 1 < R) CTI_MM_LOAD(r + 1)                      // (behind step 2: its 32 registers are the encoder's while that runs)
#endif
    }
#undef CTI_MM_LOAD
}

// =====================================================================================================
// Modes 1 + 2 + 3 in ONE kernel for FEW answer tokens (A <= 6: the FFOE / MC models, A = 3 / 6): the M tile of a rank never leaves the
// registers.  Same two MFMA steps as mbuild_mfma_kernel; instead of splitting the step-2 tile to planes and handing 528 MB of M to a GEMM
// whose N is 3 (98 % padding), every lane contracts the 16 (g, k) values it holds for its column q with the matching entries of Ar[a]
// (the sample's A x K block, staged in LDS once: A*K*4 <= 12 KiB) and keeps out[v, q, a, g] partial sums over the ranks in registers
// (ceil(V/16) object rows per wave x A x G).  The two k-halves of a lane pair meet once at the end.  fp32 FMAs on fp32-grade M values.
// =====================================================================================================
typedef short mb_s16x4 __attribute__((ext_vector_type(4)));
typedef float mb_f32x4 __attribute__((ext_vector_type(4)));
// four fp32 values -> bf16 hi + bf16 lo (the residual), as the 16x16x16 MFMA's 4-element operands
__device__ __forceinline__ void mb_split4(const float4 a, mb_s16x4& hi, mb_s16x4& lo) {
    const float x[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = static_cast<__bf16>(x[e]);
        hi[e] = __builtin_bit_cast(short, h);
        lo[e] = __builtin_bit_cast(short, static_cast<__bf16>(x[e] - static_cast<float>(h)));
    }
}
// fp32-grade product on the 16x16x16 MFMA: a * b ~= ah bh + ah bl + al bh  (TERMS == 1: the plain-bf16 mode, hi x hi only)
template <int TERMS>
__device__ __forceinline__ mb_f32x4 mb_mfma3(const mb_s16x4 ah, const mb_s16x4 al, const mb_s16x4 bh, const mb_s16x4 bl, mb_f32x4 c) {
    if (TERMS == 3) {
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c, 0, 0, 0);
}
template <int TERMS>
__device__ __forceinline__ void mb_split4t(const float4 a, mb_s16x4& hi, mb_s16x4& lo) {
    if (TERMS == 3) { mb_split4(a, hi, lo); return; }
    const float x[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) hi[e] = __builtin_bit_cast(short, static_cast<__bf16>(x[e]));
    lo = hi;                                                    // (unused)
}
template <int TERMS>
__device__ __forceinline__ void mb_split8t(const float4 a, const float4 b, mb_bf16x8& hi, mb_bf16x8& lo) {
    if (TERMS == 3) { mb_split8(a, b, hi, lo); return; }
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) hi[e] = static_cast<__bf16>(x[e]);
    lo = hi;
}

// Round 3: steps 2 AND 3 on the matrix cores, chained through registers.  The first form contracted the step-2 tile (32 x 32: rows (g, k), columns q)
// with A^ on the VALU: V*Q*A*G*h FMAs per sample with only Q of a tile's 32 column lanes doing useful work -- 60 % of the kernel's issue slots.
// Here step 2 runs per (v, g) on the 16x16x16 MFMA -- D1[k, q] = sum_j X[v, g, k, j] Q^[q, j]: rows k, columns q (12-14 of 16 useful) -- and
// its accumulator layout (column = lane & 15, rows 4 (lane >> 4) .. + 3) IS the B-operand layout of the same instruction, so mode 3,
// O[a, q] += sum_k A^[a, r, k] D1[k, q], takes D1 straight from the registers it was accumulated in (split to bf16 hi + lo: 12 VALU per tile):
// no LDS round trip, no VALU contraction, 4 accumulator registers per (v, g) for any A <= 16, and no cross-lane sum at the end.
template <int VT, int TERMS>
__global__ __launch_bounds__(1024) void mbuild_core_small_kernel(const float* __restrict__ Vr, const float* __restrict__ Qr,
                                                                 const float* __restrict__ Tt, const float* __restrict__ Ar,
                                                                 float* __restrict__ out, int V, int Q, int A, int R,
                                                                 const uint8_t* __restrict__ sm_mask, float* __restrict__ sm_p, int v_rep) {
    constexpr int HR = 16, G = 2, INNER = HR * HR * G;
    extern __shared__ __attribute__((aligned(16))) float X2[];  // [V][G][HR(k)][MB_XP], then Ar[b]: [A][K]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;  // 16 waves
    const int l31 = lane & 31, kg = lane >> 5;
    const int l15 = lane & 15, l4 = lane >> 4;                  // 16x16x16 operand roles: row / column l15, 4-deep K slice l4
    const int K = R * HR;
    float* ArS = X2 + (size_t)V * G * HR * MB_XP;
    for (int i = threadIdx.x; i < A * K; i += 1024) ArS[i] = Ar[(int64_t)b * A * K + i];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* vb = Vr + (int64_t)(b / v_rep) * V * K + kg * 8;        // v_rep > 1: rows b*v_rep .. +v_rep-1 share one image (V^ holds one block per image)
    const float* qb = Qr + ((int64_t)b * Q + (l15 < Q ? l15 : 0)) * K + l4 * 4;   // step 2's B operand: Q^[q = l15][r*16 + 4 l4 .. + 3]
    const int v0 = l31, v1 = 32 + l31;
    const bool v0ok = v0 < V, v1ok = v1 < V, qok = l15 < Q, aok = l15 < A;
    const int c1 = wid * 32 + l31;
    const int xk = l31 >> 1, xg = l31 & 1;
    mb_f32x4 acc[VT][G];                                        // O[a = 4 l4 + i, q = l15] of (v = wid + 16 t, g)
#pragma unroll
    for (int t = 0; t < VT; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[t][g] = mb_f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a00 = z4, a01 = z4, a10 = z4, a11 = z4, t0 = z4, t1 = z4, q0 = z4;
#define CTI_MC_LOAD(rr)                                                                                             \
    {                                                                                                               \
        const int o_ = (rr) * HR;                                                                                   \
        if (v0ok) { a00 = *reinterpret_cast<const float4*>(vb + (int64_t)v0 * K + o_); a01 = *reinterpret_cast<const float4*>(vb + (int64_t)v0 * K + o_ + 4); } \
        if (v1ok) { a10 = *reinterpret_cast<const float4*>(vb + (int64_t)v1 * K + o_); a11 = *reinterpret_cast<const float4*>(vb + (int64_t)v1 * K + o_ + 4); } \
        if (qok)  q0 = *reinterpret_cast<const float4*>(qb + o_);                                                   \
        const float* tp_ = Tt + ((int64_t)(rr) * INNER + c1) * HR + kg * 8;                                         \
        t0 = *reinterpret_cast<const float4*>(tp_); t1 = *reinterpret_cast<const float4*>(tp_ + 4);                 \
    }
#ifndef CTI_MC_PREFETCH
#define CTI_MC_PREFETCH 1        // rank r + 1's fragments are loaded under rank r's arithmetic, into the fp32 registers rank r's splits have just freed
#endif
    constexpr bool PF = CTI_MC_PREFETCH && VT <= 3;            // (four object rows per wave, V > 48: the prefetch registers would spill)
    if (PF) CTI_MC_LOAD(0)
    for (int r = 0; r < R; ++r) {
        if (!PF) CTI_MC_LOAD(r)
        mb_bf16x8 ah0, al0, ah1, al1, th, tl;
        mb_split8t<TERMS>(a00, a01, ah0, al0);
        mb_split8t<TERMS>(a10, a11, ah1, al1);
        mb_split8t<TERMS>(t0, t1, th, tl);
        mb_s16x4 qh, ql;
        mb_split4t<TERMS>(q0, qh, ql);
        if (PF && r + 1 < R) CTI_MC_LOAD(r + 1)
        mb_f32x16 x0, x1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { x0[e] = 0.f; x1[e] = 0.f; }
        if (TERMS == 3) {
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, th, x0, 0, 0, 0);
            x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, tl, x0, 0, 0, 0);
        }
        x0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, th, x0, 0, 0, 0);
        if (V > 32) {
            if (TERMS == 3) {
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, th, x1, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, tl, x1, 0, 0, 0);
            }
            x1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, th, x1, 0, 0, 0);
        }
        __syncthreads();                                        // step-2 readers of the previous rank are done with X2 (and Ar[b] is staged)
        mb_store_x(X2, x0, x1, V, kg, xg, xk, wid, lane);
        __syncthreads();
        // mode 3's A operand of this rank: A^[a = l15][r*16 + 4 l4 .. + 3] (zero rows beyond A), shared by every (v, g) tile of the wave
        mb_s16x4 arh, arl;
        mb_split4t<TERMS>(aok ? *reinterpret_cast<const float4*>(ArS + l15 * K + r * HR + l4 * 4) : z4, arh, arl);
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            const int v = wid + 16 * t;
            if (v < V) {                                        // (wave-uniform)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    // step 2: A operand X[v, g, k = l15, j = 4 l4 .. + 3] from LDS (16-B reads, pitch 20 floats), B operand Q^ from registers
                    mb_s16x4 xh, xl;
                    mb_split4t<TERMS>(*reinterpret_cast<const float4*>(X2 + ((v * G + g) * HR + l15) * MB_XP + l4 * 4), xh, xl);
                    const mb_f32x4 d1 = mb_mfma3<TERMS>(xh, xl, qh, ql, mb_f32x4{0.f, 0.f, 0.f, 0.f});        // D1[k = 4 l4 + i, q = l15]
                    // mode 3: D1 is already in B-operand position (k = 4 l4 + i of column q = l15)
                    mb_s16x4 dh, dl;
                    mb_split4t<TERMS>(make_float4(d1[0], d1[1], d1[2], d1[3]), dh, dl);
                    acc[t][g] = mb_mfma3<TERMS>(arh, arl, dh, dl, acc[t][g]);                                 // O[a = 4 l4 + i, q = l15]
                }
            }
        }
    }
#undef CTI_MC_LOAD
    // lane (l15 = q, l4): out[b, v, q, a = 4 l4 + i, g] for i < 4
    if (sm_p == nullptr) {
#pragma unroll
        for (int t = 0; t < VT; ++t) {
            const int v = wid + 16 * t;
            if (v < V && qok) {
                float* o = out + (((int64_t)b * V + v) * Q + l15) * A * G;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int a = 4 * l4 + i;
                    if (a < A) { o[a * G] = acc[t][0][i]; o[a * G + 1] = acc[t][1][i]; }
                }
            }
        }
        return;
    }
    // TriAttention's masked softmax (reference src/attention.py:55-58) in the same kernel: the workgroup holds ALL of the sample's V*Q*A*G
    // logits in registers, so the per-glimpse maximum and sum are two workgroup reductions and `logits` (-inf on the all-zero rows of v) and
    // `p` are written once -- no second and third pass over the logits, no extra launches.  An all-masked sample keeps the reference's NaN row.
    constexpr float L2E = 1.4426950408889634f;
    const float ninf = -__builtin_huge_valf();
    __syncthreads();                                            // every wave is done with X2 / ArS: the reductions reuse the LDS
    float* red = X2;                                            // [16 waves][2]
    bool rowok[VT], live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) live[i] = qok && 4 * l4 + i < A;
    float mx[G] = {ninf, ninf};
#pragma unroll
    for (int t = 0; t < VT; ++t) {
        const int v = wid + 16 * t;
        rowok[t] = v < V && sm_mask[(int64_t)b * V + (v < V ? v : 0)] == 0;       // (wave-uniform)
        if (rowok[t]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (live[i]) { mx[0] = fmaxf(mx[0], acc[t][0][i]); mx[1] = fmaxf(mx[1], acc[t][1][i]); }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) { const float w = wave_max(mx[g]); if (lane == 0) red[wid * 2 + g] = w; }
    __syncthreads();
    float gm[G] = {ninf, ninf};
#pragma unroll
    for (int w = 0; w < 16; ++w) { gm[0] = fmaxf(gm[0], red[w * 2]); gm[1] = fmaxf(gm[1], red[w * 2 + 1]); }
    __syncthreads();
    float e[VT][G][4];
    float sum[G] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < VT; ++t)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // exp(x - m) = exp2(x log2e - m log2e); masked rows contribute exp(-inf) = 0; m = -inf (every row masked) gives NaN like the reference
                e[t][g][i] = rowok[t] ? __builtin_amdgcn_exp2f(fmaf(acc[t][g][i], L2E, -gm[g] * L2E)) : (gm[g] == ninf ? __builtin_nanf("") : 0.f);
                if (live[i] && wid + 16 * t < V) sum[g] += e[t][g][i];
            }
#pragma unroll
    for (int g = 0; g < G; ++g) { const float w = wave_sum(sum[g]); if (lane == 0) red[wid * 2 + g] = w; }
    __syncthreads();
    float tot[G] = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 16; ++w) { tot[0] += red[w * 2]; tot[1] += red[w * 2 + 1]; }
    const float inv[G] = {1.f / tot[0], 1.f / tot[1]};
#pragma unroll
    for (int t = 0; t < VT; ++t) {
        const int v = wid + 16 * t;
        if (v < V && qok) {
            const int64_t off = (((int64_t)b * V + v) * Q + l15) * A * G;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int a = 4 * l4 + i;
                if (a < A) {
                    out[off + a * G] = rowok[t] ? acc[t][0][i] : ninf; out[off + a * G + 1] = rowok[t] ? acc[t][1][i] : ninf;
                    sm_p[off + a * G] = e[t][0][i] * inv[0]; sm_p[off + a * G + 1] = e[t][1][i] * inv[1];
                }
            }
        }
    }
}


// =====================================================================================================
// Round 6: the few-answer core with the contraction order turned round (reference src/Tensor.py:9-20 + src/tc.py:46-50 for A <= 6).
//   out[v, (q, a, g)] = sum_{r, i} V^[v, (r, i)] W[(r, i), (q, a, g)],      W[(r, i), (q, a, g)] = sum_j Q^[q, (r, j)] sum_k T[r, i, j, k, g] A^[a, (r, k)]
// The kernel above multiplies the image side in first (X = V^ T_r: 36 x 16 x 512 per rank), which makes every rank a workgroup-wide LDS exchange of X
// between two barriers: 64 barriers and ~520 instructions per wave AND RANK, of which 14 are MFMAs (142 us per 256 samples: instruction issue, not
// arithmetic).  With the few answers and the question contracted first the ranks are INDEPENDENT until one final GEMM:
//   phase A (a wave per rank, no barrier between ranks): for every (i, g) the 16 x 16 block T[r, i, g][j][k] (k contiguous in the derived layout Tk: one
//       16-B load per lane) x A^_r -> D[j, a] on the 16x16x16 MFMA, whose accumulator layout IS the B-operand layout of the next product
//       W_t[q, a] = sum_j Q^_r[q, j] D[j, a]; four consecutive i are packed per lane into one 8-B LDS store of W in the final GEMM's B-operand order;
//   phase B (one barrier): out tiles (16 v x 16 n) over K = 512, A operand = V^ rows from global memory, B operand = W from LDS.
// 5.4 M multiply-adds per sample instead of 19 M, ~1 000 instructions per wave for the WHOLE sample instead of ~16 000, 2 barriers per chunk of ranks
// (one chunk when W fits the LDS: 128 x NP x 8 B per bf16 plane, NP = G Q A rounded up to 16).
// =====================================================================================================
__global__ void tk_layout_kernel(const float* __restrict__ Teff, unsigned short* __restrict__ Tk, int64_t n) {
    // Teff[(r, i)][(j, k, g)] fp32 -> Tk: two bf16 planes (hi = bf16(x), lo = bf16(x - hi)) of [(r, i)][g][j][k]   (hr = 16, G = 2: 512 values per (r, i))
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t ri = t >> 9;
    const int c = (int)(t & 511), g = c >> 8, j = (c >> 4) & 15, k = c & 15;
    const float x = Teff[ri * 512 + ((j * 16 + k) << 1) + g];
    const unsigned short h = bf16_bits(x);
    Tk[t] = h;
    Tk[n + t] = bf16_bits(x - bf16_to_f32(h));
}

#ifndef CTI_AQ_ABL
#define CTI_AQ_ABL 0      // timing-only ablation mask: 1 no phase B, 2 no output stores, 4 no T loads, 8 no W stores, 16 no phase A at all
#endif
constexpr int AQ_NW = 8;                         // waves per workgroup: two per SIMD, 256 registers each (sixteen waves of 128 registers spilled the fragment buffers)
template <int TERMS, int MAXT>
__global__ __launch_bounds__(64 * AQ_NW) void core_small_aq_kernel(const float* __restrict__ Vr, const float* __restrict__ Qr, const unsigned short* __restrict__ Tk,
                                                             const float* __restrict__ Ar, float* __restrict__ out, int V, int Q, int A, int R, int NCH,
                                                             const uint8_t* __restrict__ sm_mask, float* __restrict__ sm_p, int v_rep) {
    constexpr int G = 2, NPL = TERMS == 3 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned long long Wimg[];   // [plane][K4 = (rank in chunk) * 4 + i / 4][n] x 8 B = four consecutive i of one (rank, n) as bf16; then 64 trash slots
    const int b = blockIdx.x;
    constexpr int NW = AQ_NW;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, s4 = lane >> 4;                                     // 16x16x16 roles: row / column c, 4-deep K slice s4
    const int K = R * 16, N = G * Q * A, NP = (N + 15) & ~15, NT = NP >> 4, VTL = (V + 15) >> 4;
    const int RC = R / NCH, K4C = RC * 4;                                        // ranks / 8-B K rows per chunk
    const int plane = K4C * NP;                                                  // (8-byte units; the whole image is < 2^15 of them)
    const int trash = NPL * plane + lane;                                     // where the lanes without a (q, a) of their own put their 8 bytes
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* vb = Vr + (int64_t)(b / v_rep) * V * K;                         // v_rep > 1: rows b*v_rep .. +v_rep-1 share one image (V^ holds one block per image)
    const bool aok = c < A, qok = c < Q;
    const int64_t tplane = (int64_t)R * 16 * 512;                                // elements per bf16 plane of Tk
    // phase-B tiles of this wave: ONE row tile vt (so that a K step's V^ fragment serves all of them), column tiles nt0, nt0 + ngrp, ...
    const int vt = wid % VTL, nt0 = wid / VTL, ngrp = (NW - vt + VTL - 1) / VTL;
    mb_f32x4 acc[MAXT];                                                          // out[v = 16 vt + 4 s4 + e, n = 16 (nt0 + u ngrp) + c]
#pragma unroll
    for (int u = 0; u < MAXT; ++u) acc[u] = mb_f32x4{0.f, 0.f, 0.f, 0.f};
    // W slots of this lane for phase A: (g, e) -> n = (g Q + 4 s4 + e) A + c, or the trash slot
    int wo[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) wo[g][e] = (aok && 4 * s4 + e < Q) ? (g * Q + 4 * s4 + e) * A + c : -1;
    typedef unsigned long long u64;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        // ---------------- phase A: W of this chunk's ranks, a wave per rank ----------------
        for (int rc = wid; rc < ((CTI_AQ_ABL & 16) ? 0 : RC); rc += NW) {
            const int r = ch * RC + rc;
            mb_s16x4 ah, al, qh, ql;
            // (columns / rows beyond A / Q read row 0: what they produce is never stored -- an exec-masked load could not be hoisted)
            mb_split4t<TERMS>(*reinterpret_cast<const float4*>(Ar + ((int64_t)b * A + (aok ? c : 0)) * K + r * 16 + s4 * 4), ah, al);   // B operand: A^[a = c][k = 4 s4 ..]
            mb_split4t<TERMS>(*reinterpret_cast<const float4*>(Qr + ((int64_t)b * Q + (qok ? c : 0)) * K + r * 16 + s4 * 4), qh, ql);   // A operand: Q^[q = c][j = 4 s4 ..]
            const u64* tr = reinterpret_cast<const u64*>(Tk + (int64_t)r * 16 * 512 + c * 16 + s4 * 4);                               // + (i * 512 + g * 256) / 4: T[r, i, g][j = c][k = 4 s4 ..]
            u64 th_[2][4][G], tl_[2][4][G];                                      // two sets: the loads of i group ig + 1 fly under the products of ig
#define CTI_AQ_LOADT(set, ig_)                                                                                       \
            _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                             \
                _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                          \
                    th_[set][ii][g] = (CTI_AQ_ABL & 4) ? (u64)(lane + ii) : tr[(((ig_) * 4 + ii) * 512 + g * 256) >> 2];       \
                    if (TERMS == 3) tl_[set][ii][g] = tr[(tplane + ((ig_) * 4 + ii) * 512 + g * 256) >> 2];              \
                }
            // i groups in pairs: group 2 p on set 0 while set 1 loads group 2 p + 1, then the other way round (a fully unrolled loop let the scheduler hoist
            // every load of the rank and spill)
#define CTI_AQ_GROUP(set, ig_)                                                                                                 \
            _Pragma("unroll") for (int g = 0; g < G; ++g) {                                                                        \
                float w[4][4];                                                                                                     \
                _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                                 \
                    const mb_s16x4 th = __builtin_bit_cast(mb_s16x4, th_[set][ii][g]);                                             \
                    const mb_s16x4 tl = TERMS == 3 ? __builtin_bit_cast(mb_s16x4, tl_[set][ii][g]) : th;                           \
                    mb_s16x4 dh, dl;                                                                                               \
                    const mb_f32x4 d = mb_mfma3<TERMS>(th, tl, ah, al, mb_f32x4{0.f, 0.f, 0.f, 0.f});     /* D[j = 4 s4 + e][a = c] */ \
                    mb_split4t<TERMS>(make_float4(d[0], d[1], d[2], d[3]), dh, dl);                       /* ... already in B-operand position (K = j) */ \
                    const mb_f32x4 wt = mb_mfma3<TERMS>(qh, ql, dh, dl, mb_f32x4{0.f, 0.f, 0.f, 0.f});    /* W_t[q = 4 s4 + e][a = c] */ \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) w[ii][e] = wt[e];                                                \
                }                                                                                                                  \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                    \
                    mb_s16x4 hi, lo;                                                                                               \
                    mb_split4t<TERMS>(make_float4(w[0][e], w[1][e], w[2][e], w[3][e]), hi, lo);                                    \
                    const int o = wo[g][e] < 0 ? trash : (rc * 4 + (ig_)) * NP + wo[g][e];                                         \
                    if (!(CTI_AQ_ABL & 8) || hi[0] == 12345) Wimg[o] = __builtin_bit_cast(u64, hi);                                \
                    if (TERMS == 3) Wimg[wo[g][e] < 0 ? trash : plane + o] = __builtin_bit_cast(u64, lo);                          \
                }                                                                                                                  \
            }
            CTI_AQ_LOADT(0, 0)
#pragma unroll 1
            for (int igp = 0; igp < 2; ++igp) {
                CTI_AQ_LOADT(1, 2 * igp + 1)
                CTI_AQ_GROUP(0, 2 * igp)
                __builtin_amdgcn_sched_barrier(0);
                if (igp == 0) CTI_AQ_LOADT(0, 2)
                CTI_AQ_GROUP(1, 2 * igp + 1)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef CTI_AQ_GROUP
#undef CTI_AQ_LOADT
        }
        if (ch == 0 && NP > N) {                                                 // the padding columns of the last n tile: zeros, once (no rank writes them)
            for (int t = threadIdx.x; t < K4C * (NP - N) * NPL; t += 64 * NW) {
                const int pl = t / (K4C * (NP - N)), rem = t - pl * (K4C * (NP - N));
                Wimg[pl * plane + (rem / (NP - N)) * NP + N + rem % (NP - N)] = 0ull;
            }
        }
        // ---------------- phase B: out tiles (16 v x 16 n) += V^[:, chunk] W[chunk, :] ----------------
        // The V^ fragments of the first sixteen K steps are in flight BEFORE the barrier (rows beyond V read row 0: their output rows are never stored), and
        // every group of eight is re-filled two groups ahead: the first form loaded one fragment per step behind an exec-masked branch -- 32 dependent L2
        // round trips, 43 of the kernel's 67 us.
        constexpr int GB = 8 / ((TERMS == 3 ? 2 : 1) * (MAXT > 3 ? 2 : 1));             // fragments per refill group (two groups in flight): what 256 registers hold beside the accumulators
        const int v = vt * 16 + c;
        const float* vrow = vb + (int64_t)(v < V ? v : 0) * K + ch * RC * 16 + s4 * 4;
        float4 vbuf[2][GB];
#define CTI_AQ_LOADV(set, k0_) _Pragma("unroll") for (int j_ = 0; j_ < GB; ++j_) vbuf[set][j_] = *reinterpret_cast<const float4*>(vrow + ((k0_) + j_) * 16);
#define CTI_AQ_STEPS(set, k0_)                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < GB; ++j_) {                                                                            \
            mb_s16x4 vh, vl;                                                                                                           \
            mb_split4t<TERMS>(vbuf[set][j_], vh, vl);                                                                                  \
            _Pragma("unroll") for (int u = 0; u < MAXT; ++u) {                                                                         \
                const int nt = nt0 + u * ngrp;                                                                                         \
                if (nt < NT) {                                               /* (wave-uniform) */                                      \
                    const mb_s16x4 bh = __builtin_bit_cast(mb_s16x4, wb_[((k0_) + j_) * 4 * NP + nt * 16]);                    \
                    const mb_s16x4 bl = TERMS == 3 ? __builtin_bit_cast(mb_s16x4, wb_[plane + ((k0_) + j_) * 4 * NP + nt * 16]) : bh; \
                    acc[u] = mb_mfma3<TERMS>(vh, vl, bh, bl, acc[u]);                                                                  \
                }                                                                                                                      \
            }                                                                                                                          \
            if ((j_ & 1) == 1) __builtin_amdgcn_sched_barrier(0);    /* (two steps' W reads in flight at most: hoisting a whole group's spills) */ \
        }
        CTI_AQ_LOADV(0, 0)
        if (RC > GB) CTI_AQ_LOADV(1, GB)
        __syncthreads();
        {
            const u64* wb_ = Wimg + s4 * NP + c;
#pragma unroll 1
            for (int k0 = 0; k0 < ((CTI_AQ_ABL & 1) ? 0 : RC); k0 += 2 * GB) {    // RC is a multiple of 8 (launcher): an even number of groups
                CTI_AQ_STEPS(0, k0)
                if (k0 + 2 * GB < RC) CTI_AQ_LOADV(0, k0 + 2 * GB)
                if (k0 + GB < RC) {
                    CTI_AQ_STEPS(1, k0 + GB)
                    if (k0 + 3 * GB < RC) CTI_AQ_LOADV(1, k0 + 3 * GB)
                }
            }
        }
#undef CTI_AQ_LOADV
#undef CTI_AQ_STEPS
        if (ch + 1 < NCH) __syncthreads();                                       // the next chunk's ranks overwrite W
    }
    // lane (c, s4) of tile (vt, nt): out[b, v = 16 vt + 4 s4 + e, q, a, g] with n = 16 nt + c = (g Q + q) A + a
    int og[MAXT]; int64_t ooff[MAXT]; bool nok[MAXT];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        const int nt = nt0 + u * ngrp, n = nt * 16 + c;
        nok[u] = nt < NT && n < N;
        const int g = n / (Q * A), rem = n - g * (Q * A), q = rem / A, a = rem - q * A;
        og[u] = g;
        ooff[u] = (((int64_t)b * V + vt * 16 + 4 * s4) * Q + q) * A * G + a * G + g;          // + e * Q * A * G
    }
    const int64_t vstride = (int64_t)Q * A * G;
    if (sm_p == nullptr) {
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            if (nok[u]) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (vt * 16 + 4 * s4 + e < V) out[ooff[u] + e * vstride] = acc[u][e];
            }
        }
        return;
    }
    // TriAttention's masked softmax (reference src/attention.py:55-58) in the same kernel, as in mbuild_core_small_kernel: the workgroup holds ALL of the
    // sample's logits, so the per-glimpse maximum and sum are two workgroup reductions; an all-masked sample keeps the reference's NaN row.
    constexpr float L2E = 1.4426950408889634f;
    const float ninf = -__builtin_huge_valf();
    __syncthreads();                                                             // every wave is done with W: the reductions reuse the LDS
    float* red = reinterpret_cast<float*>(Wimg);                                 // [NW waves][2]
    bool rowok[MAXT][4];
    float mx[G] = {ninf, ninf};
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = vt * 16 + 4 * s4 + e;
            rowok[u][e] = nt0 + u * ngrp < NT && v < V && sm_mask[(int64_t)b * V + (v < V ? v : 0)] == 0;
            if (rowok[u][e] && nok[u]) { if (og[u] == 0) mx[0] = fmaxf(mx[0], acc[u][e]); else mx[1] = fmaxf(mx[1], acc[u][e]); }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) { const float w = wave_max(mx[g]); if (lane == 0) red[wid * 2 + g] = w; }
    __syncthreads();
    float gm[G] = {ninf, ninf};
#pragma unroll
    for (int w = 0; w < NW; ++w) { gm[0] = fmaxf(gm[0], red[w * 2]); gm[1] = fmaxf(gm[1], red[w * 2 + 1]); }
    __syncthreads();
    float ex[MAXT][4];
    float sum[G] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        const float m = og[u] == 0 ? gm[0] : gm[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // exp(x - m) = exp2(x log2e - m log2e); masked rows contribute exp(-inf) = 0; m = -inf (every row masked) gives NaN like the reference
            ex[u][e] = rowok[u][e] ? __builtin_amdgcn_exp2f(fmaf(acc[u][e], L2E, -m * L2E)) : (m == ninf ? __builtin_nanf("") : 0.f);
            if (nok[u] && vt * 16 + 4 * s4 + e < V) { if (og[u] == 0) sum[0] += ex[u][e]; else sum[1] += ex[u][e]; }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) { const float w = wave_sum(sum[g]); if (lane == 0) red[wid * 2 + g] = w; }
    __syncthreads();
    float tot[G] = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) { tot[0] += red[w * 2]; tot[1] += red[w * 2 + 1]; }
    const float inv[G] = {1.f / tot[0], 1.f / tot[1]};
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        if (nok[u]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (vt * 16 + 4 * s4 + e < V) {
                    out[ooff[u] + e * vstride] = rowok[u][e] ? acc[u][e] : ninf;
                    sm_p[ooff[u] + e * vstride] = ex[u][e] * (og[u] == 0 ? inv[0] : inv[1]);
                }
            }
        }
    }
}


// =====================================================================================================
// Round 6: the direct-encoding M build without LDS and without barriers (the M build runs alone between the a side's last product and the
// mode-3 product, 0.27 ms of the headline step; reference src/Tensor.py:6-13).  mbuild_mfma_f6_kernel above exchanges X = V^ T_r through the LDS between two
// workgroup barriers for every rank and encodes on a fraction of its lanes.  Here both contractions of a (rank, k, g) are 16x16x16 MFMAs chained through
// registers, with the image side multiplied in TRANSPOSED:
//   D[j, v]   = sum_i T[r, i, j, k, g] V^[v, (r, i)]      A operand T (rows j, i contiguous in the derived layout Tj: 8-B loads), B operand V^ from global memory
//   M^t[q, v] = sum_j Q^[q, (r, j)] D[j, v]               D's accumulator layout (rows j = 4 s + e, column v) IS this product's B-operand layout
// A wave owns a (rank pair, 16 objects, glimpse) item at a time: after 2 x 16 (r, k) steps lane (v, s) holds the 32 K values of a scale block for its four
// rows (v, q = 4 s + e, g) and encodes them in registers (f6_encode_row32_regs) -- every lane with a real (v, q) encodes, nothing waits for anything else.
// =====================================================================================================
__global__ void tj_layout_kernel(const float* __restrict__ Teff, unsigned short* __restrict__ Tj, int64_t n) {
    // Teff[(r, i)][(j, k, g)] fp32 -> Tj: two bf16 planes (hi, lo) of [(r, k)][g][j][i]
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int i = (int)(t & 15), j = (int)(t >> 4) & 15, g = (int)(t >> 8) & 1, k = (int)(t >> 9) & 15;
    const int64_t r = t >> 13;
    const float x = Teff[(r * 16 + i) * 512 + ((j * 16 + k) << 1) + g];
    const unsigned short h = bf16_bits(x);
    Tj[t] = h;
    Tj[n + t] = bf16_bits(x - bf16_to_f32(h));
}

constexpr int KQ_NW = 8;
#ifndef CTI_KQ_ABL
#define CTI_KQ_ABL 0       // timing-only ablation mask: 1 no encoder / stores, 2 no MFMA chain, 4 no T loads
#endif
__global__ __launch_bounds__(64 * KQ_NW) void mbuild_f6_kq_kernel(const float* __restrict__ Vr, const float* __restrict__ Qr, const unsigned short* __restrict__ Tj,
                                                                  F6Planes P, int V, int Q, int R) {
    constexpr int G = 2;
    typedef unsigned long long u64;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, s4 = lane >> 4;
    const int K = R * 16, VTL = (V + 15) >> 4, items = (R >> 1) * G * VTL;
    extern __shared__ __attribute__((aligned(16))) char kq_lds[];                        // [wave][lane][64 B]: the H pieces on their way out
    const int64_t tplane = (int64_t)R * 16 * 512;
    const int64_t prow_b = f6_prow(P, (int64_t)b * V * Q * G);                              // the sample's first plane row (its rows are consecutive from there)
    const float* qrow = Qr + ((int64_t)b * Q + (c < Q ? c : 0)) * K + s4 * 4;                // A operand of the second product: Q^[q = c][(r, j = 4 s4 ..)]
    // items ((kb, g), vt): the waves that run together share a (rank pair, g) slice of T
    for (int item = wid; item < items; item += KQ_NW) {
        const int vt = item % VTL, kg_ = item / VTL, g = kg_ & 1, kb = kg_ >> 1;
        const int v = vt * 16 + c;
        const float* vrow = Vr + ((int64_t)b * V + (v < V ? v : 0)) * K + s4 * 4;            // B operand of the first product: V^[v = c][(r, i = 4 s4 ..)]
        float x[4][32];                                                                     // row (v, q = 4 s4 + e, g): K = 32 kb .. + 31
        // the item's 2 x 16 (rank, k) steps as ONE stream of T fragments through a rolling window of eight (hi + lo: 32 registers): the loads of step t + 8 fly
        // under the products of step t (the first form loaded a rank's sixteen at once and waited: ~58 of the kernel's 223 us)
        const u64* tr0 = reinterpret_cast<const u64*>(Tj + ((int64_t)(2 * kb) * 16 * 2 + g) * 256 + c * 16 + s4 * 4);       // step t = 16 rl + k: + t * 512 / 4
        mb_s16x4 vh[2], vl[2], qh[2], ql[2];
#pragma unroll
        for (int rl = 0; rl < 2; ++rl) {
            mb_split4(*reinterpret_cast<const float4*>(vrow + (2 * kb + rl) * 16), vh[rl], vl[rl]);
            mb_split4(*reinterpret_cast<const float4*>(qrow + (2 * kb + rl) * 16), qh[rl], ql[rl]);
        }
        u64 th_[8], tl_[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { th_[t] = (CTI_KQ_ABL & 4) ? (u64)(lane + t) : tr0[t * 128]; tl_[t] = (CTI_KQ_ABL & 4) ? (u64)(lane * 3 + t) : tr0[(tplane >> 2) + t * 128]; }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int rl = t >> 4;
            const mb_s16x4 th = __builtin_bit_cast(mb_s16x4, th_[t & 7]), tl = __builtin_bit_cast(mb_s16x4, tl_[t & 7]);
            if (t + 8 < 32) { th_[t & 7] = (CTI_KQ_ABL & 4) ? (u64)(lane + t) : tr0[(t + 8) * 128]; tl_[t & 7] = (CTI_KQ_ABL & 4) ? (u64)(lane * 3 + t) : tr0[(tplane >> 2) + (t + 8) * 128]; }
            mb_s16x4 dh, dl;
            if (CTI_KQ_ABL & 2) { x[0][t] = (float)th[0] + (float)tl[1]; x[1][t] = (float)vh[rl][0]; x[2][t] = (float)qh[rl][1]; x[3][t] = (float)ql[rl][2] + (float)vl[rl][3]; continue; }
            const mb_f32x4 d = mb_mfma3<3>(th, tl, vh[rl], vl[rl], mb_f32x4{0.f, 0.f, 0.f, 0.f});          // D[j = 4 s4 + e][v = c]
            mb_split4(make_float4(d[0], d[1], d[2], d[3]), dh, dl);
            const mb_f32x4 mt = mb_mfma3<3>(qh[rl], ql[rl], dh, dl, mb_f32x4{0.f, 0.f, 0.f, 0.f});         // M^t[q = 4 s4 + e][v = c]
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e][t] = mt[e];
        }
        // encode: lane (c, s4) owns rows (v, q = 4 s4 + e, g).  The H piece stores go through a wave-private LDS patch [lane][64 B] so that FOUR lanes store one
        // row's 64 B: a lane storing its own row's four pieces makes each store instruction touch 64 cache lines (the stores were ~half of the kernel)
        f6_lds_u32x4* const hl = (f6_lds_u32x4*)((__attribute__((address_space(3))) char*)kq_lds + wid * 4096 + lane * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = 4 * s4 + e;
            if (v < V && q < Q && (!(CTI_KQ_ABL & 1) || x[0][0] == 12345.f)) {
                const int64_t prow = prow_b + (v * Q + q) * G + g;
                const int64_t o = (int64_t)kb * P.rows_alloc + prow;
                f6_encode_row32_regs<false>(x[e], -__builtin_huge_valf(), nullptr, reinterpret_cast<char*>(P.FL) + o * 24,
                                            reinterpret_cast<char*>(P.S) + ((int64_t)kb * P.rows_allocS + prow) * 2, 1, hl);
            }
            // (same wave writes and reads the patch: LDS operations of one wave complete in order; the compiler keeps the order across the clobbers)
            asm volatile("" ::: "memory");
            if (!(CTI_KQ_ABL & 1)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int slot = 16 * j + (lane >> 2), piece = lane & 3;              // slot = the lane that encoded: (c' = slot & 15, s4' = slot >> 4) = (slot & 15, j)
                    const int v2 = vt * 16 + (slot & 15), q2 = 4 * j + e;
                    if (v2 < V && q2 < Q) {
                        const f6_u32x4 d = *(f6_lds_u32x4*)((__attribute__((address_space(3))) char*)kq_lds + wid * 4096 + slot * 64 + piece * 16);
                        const int64_t o2 = (int64_t)kb * P.rows_alloc + prow_b + (v2 * Q + q2) * G + g;
                        *reinterpret_cast<f6_u32x4*>(reinterpret_cast<char*>(P.H) + o2 * 64 + piece * 16) = d;
                    }
                }
            }
            asm volatile("" ::: "memory");
        }
    }
}

}  // namespace

bool mbuild_mfma_fits(int B, int V, int Q, int R, int hr, int G) {
#ifdef CTI_NO_MBUILD_MFMA
    return false;
#endif
    return hr == 16 && G == 2 && V >= 1 && V <= 64 && Q >= 1 && Q <= 16 && (R & 1) == 0 && B <= 65535 &&
           sizeof(float) * ((size_t)V * G * 16 * MB_XP + 16 * 16 * MB_SP) <= 160 * 1024;
}

// Tt: the core pre-transposed to [r][c][i] (cti_transpose_f32 of T_eff[r] (i x c) for every r).  CTI_E_UNSUPPORTED = take mbuild_fast.
int mbuild_mfma(const float* Vr, const float* Qr, const float* Tt, unsigned short* Mh, unsigned short* Ml, float* Mf, int B, int V, int Q, int R,
                int hr, int G, int64_t pitchM, hipStream_t st) {
    if (!mbuild_mfma_fits(B, V, Q, R, hr, G) || !Tt || !((Mh && Ml) || Mf)) return CTI_E_UNSUPPORTED;
    if (Mf && ((reinterpret_cast<uintptr_t>(Mf) & 15) || (pitchM & 3))) return CTI_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Vr) | reinterpret_cast<uintptr_t>(Qr) | reinterpret_cast<uintptr_t>(Tt)) & 15) return CTI_E_UNSUPPORTED;
    const size_t lds = sizeof(float) * ((size_t)V * G * 16 * MB_XP + 16 * 16 * MB_SP);
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mbuild_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mbuild_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail((int)e, "mbuild_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    if (Mf) hipLaunchKernelGGL(mbuild_mfma_kernel<true>, dim3(B), dim3(1024), lds, st, Vr, Qr, Tt, Mh, Ml, Mf, V, Q, R, pitchM);
    else    hipLaunchKernelGGL(mbuild_mfma_kernel<false>, dim3(B), dim3(1024), lds, st, Vr, Qr, Tt, Mh, Ml, Mf, V, Q, R, pitchM);
    return launch_status("mbuild_mfma");
}

// The shape tests of the two fused kernels, by sizes only: cti_tcnet.hip plans its workspace with the SAME predicates the launchers apply
// (a plan that accepts a shape its kernel refuses leaves the forward without M / A^ planes to fall back on).
static size_t mbuild_mfma_f6_lds(int V, int Q, int G) { return sizeof(float) * ((size_t)V * G * 16 * MB_XP + (size_t)V * 4 * (2 * Q) * 4); }
static size_t mbuild_core_small_lds(int V, int A, int R, int hr, int G) { return sizeof(float) * ((size_t)V * G * 16 * MB_XP + (size_t)A * R * hr); }
bool mbuild_mfma_f6_fits(int B, int V, int Q, int R, int hr, int G) {
#if defined(CTI_NO_MBUILD_MFMA) || defined(CTI_NO_MBUILD_F6)
    return false;
#endif
    return hr == 16 && G == 2 && V >= 1 && V <= 64 && Q >= 1 && Q <= 16 && (R & 1) == 0 && B <= 65535 && mbuild_mfma_f6_lds(V, Q, G) <= 160 * 1024;
}
bool mbuild_core_small_fits(int B, int V, int Q, int A, int R, int hr, int G) {
#ifdef CTI_NO_MBUILD_CORE_SMALL
    return false;
#endif
    return hr == 16 && G == 2 && V >= 1 && V <= 64 && Q >= 1 && Q <= 16 && A >= 1 && A <= 6 && (R & 1) == 0 && B <= 65535 &&
           mbuild_core_small_lds(V, A, R, hr, G) <= 160 * 1024;
}

// M straight into the f16f6 planes P of the mode-3 product (rows (b, v, q, g), K = R * 16).  CTI_E_UNSUPPORTED (no message) outside hr = 16, G = 2,
// even R and the LDS budget: the caller takes mbuild_mfma -> fp32 rows -> quantize_f16f6.
// Tj: T_eff as two bf16 planes of [(r, k)][g][j][i] for the round-6 kernel (mbuild_f6_kq_kernel); R * 16 * 512 * 4 bytes
int mbuild_f6_tj_layout(const float* Teff, float* Tj, int R, int hr, int G, hipStream_t st) {
    if (hr != 16 || G != 2) return CTI_OK;
    const int64_t n = (int64_t)R * 16 * 512;
    hipLaunchKernelGGL(tj_layout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Teff, reinterpret_cast<unsigned short*>(Tj), n);
    return launch_status("mbuild_f6_tj_layout");
}

int mbuild_mfma_f6(const float* Vr, const float* Qr, const float* Tt, const F6Planes& P, int B, int V, int Q, int R, int hr, int G, hipStream_t st, const float* Tj) {
#if defined(CTI_NO_MBUILD_MFMA) || defined(CTI_NO_MBUILD_F6)
    return CTI_E_UNSUPPORTED;
#endif
    if (!mbuild_mfma_f6_fits(B, V, Q, R, hr, G) || !Tt || P.Kb * 32 != R * hr) return CTI_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Vr) | reinterpret_cast<uintptr_t>(Qr) | reinterpret_cast<uintptr_t>(Tt)) & 15) return CTI_E_UNSUPPORTED;
    static const bool old_env = [] { const char* e = getenv("CTI_MBF6_OLD"); return e && e[0] == '1'; }();
    if (Tj && !old_env && !(reinterpret_cast<uintptr_t>(Tj) & 15)) {        // round 6: no LDS, no barriers (CTI_MBF6_OLD=1: the round-2 kernel, A/B)
        hipLaunchKernelGGL(mbuild_f6_kq_kernel, dim3(B), dim3(64 * KQ_NW), KQ_NW * 4096, st, Vr, Qr, reinterpret_cast<const unsigned short*>(Tj), P, V, Q, R);
        return launch_status("mbuild_f6_kq");
    }
    const size_t lds = mbuild_mfma_f6_lds(V, Q, G);
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mbuild_mfma_f6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail((int)e, "mbuild_mfma_f6: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    hipLaunchKernelGGL(mbuild_mfma_f6_kernel, dim3(B), dim3(1024), lds, st, Vr, Qr, Tt, P, V, Q, R);
    return launch_status("mbuild_mfma_f6");
}

// Modes 1 + 2 + 3 for few answer tokens, out (B,V,Q,A,G) fp32.  CTI_E_UNSUPPORTED (no message) outside hr = 16, G = 2, V <= 64, Q <= 16, A <= 6:
// the caller takes the M build + GEMM pair.  sm_p != NULL (with sm_mask = the zero-row mask of v): TriAttention's masked softmax in the same
// kernel -- `out` gets -inf on masked rows, sm_p the attention map.
// Tk != NULL: the round-6 kernel (answers and question contracted first, core_small_aq_kernel; Tk = T_eff as [(r, i)][g][j][k], core_small_tk_layout) where its W
// image fits the LDS in at most 8 chunks of ranks and R is a multiple of the chunk count; CTI_CORE_SMALL_OLD=1 keeps the round-3 kernel (A/B).
int core_small_tk_layout(const float* Teff, float* Tk, int R, int hr, int G, hipStream_t st) {          // Tk: R * 16 * 512 * 4 bytes = two bf16 planes
    if (hr != 16 || G != 2) return CTI_OK;                       // (the fused few-answer kernels exist for hr = 16, G = 2 only: nothing reads Tk otherwise)
    const int64_t n = (int64_t)R * 16 * 512;
    hipLaunchKernelGGL(tk_layout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Teff, reinterpret_cast<unsigned short*>(Tk), n);
    return launch_status("core_small_tk_layout");
}

static int core_small_aq_chunks(int V, int Q, int A, int R, int terms) {
    const size_t NP = (size_t)((2 * Q * A + 15) & ~15), npl = terms == 3 ? 2 : 1;
    for (int nch = 1; nch <= 4; nch *= 2)
        if (R % (8 * nch) == 0 && npl * (size_t)(R / nch) * 4 * NP * 8 + 512 <= 160 * 1024) return nch;     // (a chunk's K steps go in groups of eight)
    return 0;
}

int mbuild_core_small(const float* Vr, const float* Qr, const float* Tt, const float* Ar, float* out, int B, int V, int Q, int A, int R, int hr, int G,
                      hipStream_t st, const uint8_t* sm_mask, float* sm_p, int v_rep, int terms, const float* Tk) {
#ifdef CTI_NO_MBUILD_CORE_SMALL
    return CTI_E_UNSUPPORTED;
#endif
    if (!mbuild_core_small_fits(B, V, Q, A, R, hr, G) || !Tt) return CTI_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Vr) | reinterpret_cast<uintptr_t>(Qr) | reinterpret_cast<uintptr_t>(Tt) | reinterpret_cast<uintptr_t>(Ar)) & 15) return CTI_E_UNSUPPORTED;
    static const bool old_env = [] { const char* e = getenv("CTI_CORE_SMALL_OLD"); return e && e[0] == '1'; }();
    const int nch = (Tk && !old_env && !(reinterpret_cast<uintptr_t>(Tk) & 15)) ? core_small_aq_chunks(V, Q, A, R, terms) : 0;
    if (nch > 0) {
        const int NP = (2 * Q * A + 15) & ~15, vtl = (V + 15) / 16, maxt = (NP / 16 + AQ_NW / vtl - 1) / (AQ_NW / vtl);     // a row tile's column tiles over its 8 / vtl (or more) waves
        const size_t lds_aq = (size_t)(terms == 3 ? 2 : 1) * (R / nch) * 4 * NP * 8 + 512;
#define CTI_AQ_LAUNCH(TRM, MT)                                                                                                              \
        {                                                                                                                                   \
            auto kern = core_small_aq_kernel<TRM, MT>;                                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return fail((int)e, "mbuild_core_small: hipFuncSetAttribute: %s", hipGetErrorString(e));                   \
            hipLaunchKernelGGL(kern, dim3(B), dim3(64 * AQ_NW), lds_aq, st, Vr, Qr, reinterpret_cast<const unsigned short*>(Tk), Ar, out, V, Q, A, R, nch, sm_mask, sm_p, v_rep > 0 ? v_rep : 1); \
        }
        // (three-product mode with more than three tiles per wave -- V > 48 -- spills a register to scratch: that shape stays on the round-3 kernel;
        // tests/test_abi.py keeps every kernel that may run beside another stream scratch-free)
        if (maxt <= 3 || (terms == 1 && maxt <= 6)) {
            if (terms == 1) { if (maxt <= 3) CTI_AQ_LAUNCH(1, 3) else CTI_AQ_LAUNCH(1, 6) }
            else            CTI_AQ_LAUNCH(3, 3)
            return launch_status("core_small_aq");
        }
#undef CTI_AQ_LAUNCH
    }
    const size_t lds = mbuild_core_small_lds(V, A, R, hr, G);
    const int VT = (V + 15) / 16;
#define CTI_MC_LAUNCH(VTv, TRM)                                                                                                             \
    {                                                                                                                                       \
        auto kern = mbuild_core_small_kernel<VTv, TRM>;                                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
        if (e != hipSuccess) return fail((int)e, "mbuild_core_small: hipFuncSetAttribute: %s", hipGetErrorString(e));                       \
        hipLaunchKernelGGL(kern, dim3(B), dim3(1024), lds, st, Vr, Qr, Tt, Ar, out, V, Q, A, R, sm_mask, sm_p, v_rep > 0 ? v_rep : 1);      \
    }
    // terms = 1: the plain-bf16 mode (CTI_PREC_BF16: one product per pair, no lo parts -- the hi / lo splits are this kernel's VALU bound)
    if (terms == 1) { if (VT <= 3) CTI_MC_LAUNCH(3, 1) else CTI_MC_LAUNCH(4, 1) }
    else            { if (VT <= 3) CTI_MC_LAUNCH(3, 3) else CTI_MC_LAUNCH(4, 3) }
#undef CTI_MC_LAUNCH
    return launch_status("mbuild_core_small");
}

}  // namespace cti
