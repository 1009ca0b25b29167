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

namespace cti {
namespace {

__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x)); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

constexpr int MB_ITEMS = 4;      // step-1 work items per thread the fast path supports (HR^2*G*ceil(V/8) <= 4096)
constexpr int MB_ROWS = 2;       // output rows per thread (V*Q*G <= 2048)

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
    const int Qpad = Q | 1;
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
    // step-1 items: (column c of T_eff, chunk of 8 v)
    const int nvc = (V + 7) / 8, nitems = inner * nvc;
    int it_c[MB_ITEMS], it_v0[MB_ITEMS], it_x[MB_ITEMS];
#pragma unroll
    for (int n = 0; n < MB_ITEMS; ++n) {
        const int it = t + n * nthr;
        const int c = it % inner, v0 = (it / inner) * 8;
        const int g = c % G, k = (c / G) % HR, j = c / (G * HR);
        it_c[n] = it < nitems ? c : -1;
        it_v0[n] = v0;
        it_x[n] = (v0 * G + g) * HH + j * HR + k;              // X offset of (v0, g, j, k); +G*HH per v
    }
    // step-2 rows: lane order (v, g, q), q fastest: a 16-lane group shares its X row (LDS broadcast)
    int row_x[MB_ROWS], row_q[MB_ROWS];
    int64_t row_o[MB_ROWS];
#pragma unroll
    for (int n = 0; n < MB_ROWS; ++n) {
        const int lr = t + n * nthr;
        const int lq = lr % Q, lg = (lr / Q) % G, lv = lr / (Q * G);
        row_q[n] = lr < rows ? lq : -1;
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
        // step 1:  X[v][g][j][k] = sum_i T_eff[r][i][j,k,g] * Vr[v][i]
#pragma unroll
        for (int n = 0; n < MB_ITEMS; ++n) {
            const int c = it_c[n];
            if (c >= 0) {
            const int v0 = it_v0[n];
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = 0.f;
#pragma unroll 1
            for (int i4 = 0; i4 < HR; i4 += 4) {
                const float t0 = Ts[(i4 + 0) * inner + c], t1 = Ts[(i4 + 1) * inner + c];
                const float t2 = Ts[(i4 + 2) * inner + c], t3 = Ts[(i4 + 3) * inner + c];
                const float* vrow = Vs + v0 * HR + i4;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    // rows v0+u >= V read stale-but-in-bounds LDS (Vs is followed by Qs); their x[u] is never stored
                    const float4 vv = *reinterpret_cast<const float4*>(vrow + u * HR);
                    x[u] = fmaf(t0, vv.x, fmaf(t1, vv.y, fmaf(t2, vv.z, fmaf(t3, vv.w, x[u]))));
                }
            }
            float* xo = Xs + it_x[n];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v0 + u < V) xo[u * G * HH] = x[u];
            }
        }
        __syncthreads();
        // step 2: one output row (v,q,g) per thread and trip, HR columns
#pragma unroll
        for (int n = 0; n < MB_ROWS; ++n) {
            const int lq = row_q[n];
            if (lq >= 0) {
            float acc[HR];
#pragma unroll
            for (int k = 0; k < HR; ++k) acc[k] = 0.f;
            const float* xr = Xs + row_x[n];
#pragma unroll 2
            for (int j = 0; j < HR; ++j) {
                const float qv = Qs[j * Qpad + lq];
#pragma unroll
                for (int k4 = 0; k4 < HR; k4 += 4) {
                    const float4 xx = *reinterpret_cast<const float4*>(xr + j * HR + k4);
                    acc[k4 + 0] = fmaf(qv, xx.x, acc[k4 + 0]);
                    acc[k4 + 1] = fmaf(qv, xx.y, acc[k4 + 1]);
                    acc[k4 + 2] = fmaf(qv, xx.z, acc[k4 + 2]);
                    acc[k4 + 3] = fmaf(qv, xx.w, acc[k4 + 3]);
                }
            }
            const int64_t orow = row_o[n];
            const int c0 = r * HR;
            if (PLANES) {
                // chunk-major planes: column c of row orow lives at (c >> 4) * pitch + orow * 16 + (c & 15); the rows of one
                // v are 32 B apart, so a wave's stores fall in a few contiguous KiB
                unsigned short hb[HR], lb[HR];
#pragma unroll
                for (int k = 0; k < HR; ++k) { hb[k] = bf16_bits(acc[k]); lb[k] = bf16_bits(acc[k] - bf16_to_f32(hb[k])); }
                const int64_t o = (int64_t)(c0 >> 4) * ldm + orow * 16 + (c0 & 15);
                unsigned short* ph = Mh + o;
                unsigned short* pl = Ml + o;
                if (HR % 8 == 0) {
#pragma unroll
                    for (int c8 = 0; c8 < HR / 8; ++c8) {
                        const unsigned short* hh = hb + c8 * 8; const unsigned short* ll = lb + c8 * 8;
                        *reinterpret_cast<uint4*>(ph + c8 * 8) = make_uint4(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16), hh[4] | ((unsigned)hh[5] << 16), hh[6] | ((unsigned)hh[7] << 16));
                        *reinterpret_cast<uint4*>(pl + c8 * 8) = make_uint4(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16), ll[4] | ((unsigned)ll[5] << 16), ll[6] | ((unsigned)ll[7] << 16));
                    }
                } else {                                    // HR == 4: 8-byte pieces
                    *reinterpret_cast<uint2*>(ph) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
                    *reinterpret_cast<uint2*>(pl) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
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
                    for (int k4 = 0; k4 < HR; k4 += 4) *reinterpret_cast<float4*>(pf + k4) = make_float4(acc[k4], acc[k4 + 1], acc[k4 + 2], acc[k4 + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < HR; ++k) pf[k] = acc[k];
                }
            }
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
int mbuild_fast(const float* Vr, const float* Qr, const float* Teff, float* Mf, unsigned short* Mh, unsigned short* Ml, int B,
                int V, int Q, int R, int hr, int G, int64_t ldm, hipStream_t st) {
    if (hr != 4 && hr != 8 && hr != 16) return CTI_E_UNSUPPORTED;
    if (B > 65535) return CTI_E_UNSUPPORTED;
    const size_t lds = sizeof(float) * ((size_t)hr * hr * hr * G + (size_t)V * G * hr * hr + (size_t)(V + 8) * hr + (size_t)hr * (Q | 1));
    if (lds > 160 * 1024) return CTI_E_UNSUPPORTED;
    if ((int64_t)hr * hr * G * ((V + 7) / 8) > 4096 || (int64_t)V * Q * G > 2048 || hr * hr * hr * G / 4 > 4096 || V * hr > 1024 || Q * hr > 1024)
        return CTI_E_UNSUPPORTED;                           // per-thread item / row / prefetch budgets of the fast kernel
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
