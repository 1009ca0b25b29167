// cti_backward2.hip -- backward kernels of the non-GEMM ops: M build (modes 1+2), masked softmax (Tri / Bi), weighted
// sum-pools (tri / bi) and bilinear attention logits.  fp32 VALU, lane axis = the contiguous axis, LDS column
// accumulators (each thread owns its column: no synchronisation), wave-shuffle reductions where an output sums over d.
#include "cti_common.h"

namespace cti {
namespace {

// =====================================================================================================================
// M build backward.  Forward: X[v,j,k,g] = sum_i T[i,j,k,g] Vr[v,i];  M[v,q,g,k] = sum_j X[v,j,k,g] Qr[q,j]  (per b, r).
//   dQr[q,j]   = sum_{v,g,k} dM[v,q,g,k] X[v,j,k,g]
//   dX[v,j,k,g]= sum_q dM[v,q,g,k] Qr[q,j]
//   dVr[v,i]   = sum_{j,k,g} dX[v,j,k,g] T[i,j,k,g]
//   dT_b[i,j,k,g] = sum_v Vr[v,i] dX[v,j,k,g]        (per-sample partial; summed over b afterwards)
// One 1024-thread workgroup per (b, rank group); LDS: T[r] | X then dX (same buffer) | Vr slice | Qr slice.
// =====================================================================================================================
template <int HR>
__global__ __launch_bounds__(1024) void mbuild_bwd_kernel(const float* __restrict__ dM, const float* __restrict__ Vr,
                                                          const float* __restrict__ Qr, const float* __restrict__ Teff,
                                                          float* __restrict__ dVr, float* __restrict__ dQr, float* __restrict__ dTpart,
                                                          int V, int Q, int R, int G, int rpb) {
    constexpr int HH = HR * HR;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int inner = HH * G;
    float* Ts = sm;                              // [HR][inner]  (i, (j,k,g))
    float* Xs = Ts + HR * inner;                 // [V][G][HR(j)][HR(k)]   X, later dX
    float* Vs = Xs + (size_t)V * G * HH;         // [V + 8][HR]
    float* Qs = Vs + (V + 8) * HR;               // [Q][HR]  (q, j)
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    constexpr int nthr = 1024;
    const int b = blockIdx.y;
    const int K = R * HR;
    const float* vb = Vr + (int64_t)b * V * K;
    const float* qb = Qr + (int64_t)b * Q * K;
    const float* dmb = dM + (int64_t)b * V * Q * G * K;
    const int r_lo = blockIdx.x * rpb, r_hi = min(R, r_lo + rpb);
    // rank-independent index decoding, done once (runtime integer divisions cost ~40 instructions each)
    const int xc = t % inner, xsp = t / inner, xns = max(1, nthr / inner);            // X: column xc, v = xsp, xsp + xns, ...
    const int xg = xc % G, xk = (xc / G) % HR, xj = xc / (G * HR);
    const int x_off = xg * HH + xj * HR + xk;
    const int nout = Q * HR;
    int P = 1;
    while (P * 2 * nout <= nthr && P < 16) P *= 2;
    const int qo = t / P, qpart = t % P, qq = qo / HR, qj = qo % HR;                   // dQr: output (qq, qj), v split P ways
    int dx_v[2], dx_g[2], dx_k[2];                                                      // dX items (v, g, k): <= 2 per thread
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int it = t + n * nthr;
        dx_k[n] = it % HR; dx_g[n] = (it / HR) % G; dx_v[n] = it < V * G * HR ? it / (HR * G) : -1;
    }
    const int tc = t % inner, ti0 = t / inner, tistep = max(1, nthr / inner);          // dT: column tc, rows ti0, ti0 + tistep, ...
    const int tg = tc % G, tjk = tc / G;
    for (int r = r_lo; r < r_hi; ++r) {
        __syncthreads();
        const float* Tr = Teff + (int64_t)r * HR * inner;
        for (int e = t; e < HR * inner; e += nthr) Ts[e] = Tr[e];
        for (int e = t; e < V * HR; e += nthr) Vs[e] = vb[(int64_t)(e / HR) * K + r * HR + (e % HR)];
        for (int e = t; e < Q * HR; e += nthr) Qs[e] = qb[(int64_t)(e / HR) * K + r * HR + (e % HR)];
        __syncthreads();
        // X[v][g][j][k] (as the forward's step 1): the thread's T_eff column in registers, v strided over the column groups
        if (xsp < xns) {
            float tcol[HR];
#pragma unroll
            for (int i = 0; i < HR; ++i) tcol[i] = Ts[i * inner + xc];
            for (int v = xsp; v < V; v += xns) {
                float x = 0.f;
#pragma unroll
                for (int i = 0; i < HR; ++i) x = fmaf(tcol[i], Vs[v * HR + i], x);
                Xs[v * G * HH + x_off] = x;
            }
        }
        __syncthreads();
        // dQr[q][j] = sum_{v,g} sum_k dM[v,q,g,k] X[v,g,j,k]: P lanes split v, shuffle-reduced
        {
            if (t < nout * P) {
                const int part = qpart, q = qq, j = qj;
                float s = 0.f;
                for (int v = part; v < V; v += P)
                    for (int g = 0; g < G; ++g) {
                        const float* dm = dmb + (((int64_t)v * Q + q) * G + g) * K + r * HR;
                        const float* xr = Xs + (v * G + g) * HH + j * HR;
#pragma unroll
                        for (int k = 0; k < HR; ++k) s = fmaf(dm[k], xr[k], s);
                    }
                for (int off = P >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                if (part == 0) dQr[((int64_t)b * Q + q) * K + r * HR + j] = s;
            }
        }
        __syncthreads();
        // dX[v][g][j][k] = sum_q dM[v,q,g,k] Qr[q,j]  (overwrites X)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int k = dx_k[n], g = dx_g[n], v = dx_v[n];
            if (v < 0) continue;
            float acc[HR];
#pragma unroll
            for (int j = 0; j < HR; ++j) acc[j] = 0.f;
            for (int q = 0; q < Q; ++q) {
                const float m = dmb[(((int64_t)v * Q + q) * G + g) * K + r * HR + k];
#pragma unroll
                for (int j = 0; j < HR; ++j) acc[j] = fmaf(m, Qs[q * HR + j], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < HR; ++j) Xs[(v * G + g) * HH + j * HR + k] = acc[j];
        }
        __syncthreads();
        // dVr[v][i] = sum_{g,jk} dX[v][g][jk] T[i][(jk)*G + g]: one wave per (v,i) pair, lanes over (g,jk)
        for (int pr = wid; pr < V * HR; pr += nthr / 64) {
            const int v = pr / HR, i = pr % HR;
            float s = 0.f;
            for (int e = lane; e < inner; e += 64) {
                const int g = e / HH, jk = e % HH;
                s = fmaf(Xs[(v * G + g) * HH + jk], Ts[i * inner + jk * G + g], s);
            }
            s = wave_sum(s);
            if (lane == 0) dVr[((int64_t)b * V + v) * K + r * HR + i] = s;
        }
        // dT_b[r][i][c] = sum_v Vr[v][i] dX[v][c]
        for (int i = ti0; i < HR; i += tistep) {
            if (ti0 >= tistep) break;
            float s = 0.f;
            for (int v = 0; v < V; ++v) s = fmaf(Vs[v * HR + i], Xs[(v * G + tg) * HH + tjk], s);
            dTpart[((int64_t)b * R + r) * HR * inner + i * inner + tc] = s;
        }
    }
}

// Staged variant (the shapes of every model configuration: hr = 16, G = 2, V*Q*G <= 1024).  The generic kernel above reads its
// dM slice straight from global memory inside the (v,g) and q loops -- short dependent loads at one workgroup per CU, i.e. pure
// latency (2.7 ms at B = 256).  Here the rank's dM slice (V*Q*G rows x hr, 64 KiB) and T_eff[r] (32 KiB) are prefetched into
// registers one rank ahead with 16-B loads and time-multiplexed through ONE LDS region: T for the X phase, the dM slice for the
// dQr / dX phases (row pitch hr + 4 floats: the v-split lanes of dQr land in different banks), T again (still in registers) for
// the dVr / dT phases.
#ifndef CTI_MBB_SKIP
#define CTI_MBB_SKIP 0        // timing-only ablation mask (tools/tune_mbuild.py): 1 X, 2 dQr, 4 dX, 8 dVr, 16 dT
#endif
template <int HR>
__global__ __launch_bounds__(1024) void mbuild_bwd_staged_kernel(const float* __restrict__ dM, const float* __restrict__ Vr,
                                                                 const float* __restrict__ Qr, const float* __restrict__ Teff,
                                                                 float* __restrict__ dVr, float* __restrict__ dQr, float* __restrict__ dTpart,
                                                                 int V, int Q, int R, int G, int rpb, int region_floats) {
    constexpr int HH = HR * HR, HP = HR + 4, F4 = HR / 4;
    const int TP = HH * G + 8;                   // row pitch of T in the region: rows 4 apart shift by 128 B (dVr phase: conflict-free)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int inner = HH * G;
    float* Rg = sm;                              // T[r] ([HR][inner]) or the dM slice ([V*Q*G][HP])
    float* Xs = Rg + region_floats;              // [V][G][HR(j)][HR(k)]   X, later dX
    float* Vs = Xs + (size_t)V * G * HH;         // [V + 8][HR]
    float* Qs = Vs + (V + 8) * HR;               // [Q][HR]  (q, j)
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    constexpr int nthr = 1024;
    const int b = blockIdx.y;
    const int K = R * HR;
    const int rowsM = V * Q * G;
    const float* vb = Vr + (int64_t)b * V * K;
    const float* qb = Qr + (int64_t)b * Q * K;
    const float* dmb = dM + (int64_t)b * rowsM * K;
    const int r_lo = blockIdx.x * rpb, r_hi = min(R, r_lo + rpb);
    const int xc = t % inner, xsp = t / inner, xns = max(1, nthr / inner);
    const int xg = xc % G, xk = (xc / G) % HR, xj = xc / (G * HR);
    const int x_off = xg * HH + xj * HR + xk;
    const int nout = Q * HR;
    int P = 1;
    while (P * 2 * nout <= nthr && P < 16) P *= 2;
    const int qo = t / P, qpart = t % P, qq = qo / HR, qj = qo % HR;
    int dx_v[2], dx_g[2], dx_k[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int it = t + n * nthr;
        dx_k[n] = it % HR; dx_g[n] = (it / HR) % G; dx_v[n] = it < V * G * HR ? it / (HR * G) : -1;
    }
    int tistep = 1;                                                               // column groups of the dT phase: a power of two dividing HR
    while (tistep * 2 <= HR && tistep * 2 * inner <= nthr) tistep *= 2;
    const int tc = t % inner, ti0 = t / inner;
    const int tg = tc % G, tjk = tc / G;
    // prefetch registers: T[r] = HR*inner/4 float4 (<= 2 per thread), dM slice = rowsM*F4 float4 (<= 4 per thread)
    const int t4n = HR * inner / 4, d4n = rowsM * F4;
    int64_t d_off[4]; int d_dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int f = t + u * nthr;
        const int row = f / F4, part = f % F4;
        d_off[u] = f < d4n ? (int64_t)row * K + part * 4 : -1;
        d_dst[u] = row * HP + part * 4;
    }
    const int tput0 = ((4 * t) / inner) * TP + (4 * t) % inner, tput1 = ((4 * (t + nthr)) / inner) * TP + (4 * (t + nthr)) % inner;
    float4 tp0 = make_float4(0.f, 0.f, 0.f, 0.f), tp1 = tp0, dp0 = tp0, dp1 = tp0, dp2 = tp0, dp3 = tp0;
#define CTI_MBB_PF_T(rr)                                                                              \
    {                                                                                                 \
        const float4* Tr_ = reinterpret_cast<const float4*>(Teff + (int64_t)(rr) * HR * inner);       \
        if (t < t4n) tp0 = Tr_[t];                                                                    \
        if (t + nthr < t4n) tp1 = Tr_[t + nthr];                                                      \
    }
#define CTI_MBB_PF_D(rr)                                                                              \
    {                                                                                                 \
        const float* s_ = dmb + (rr) * HR;                                                            \
        if (d_off[0] >= 0) dp0 = *reinterpret_cast<const float4*>(s_ + d_off[0]);                     \
        if (d_off[1] >= 0) dp1 = *reinterpret_cast<const float4*>(s_ + d_off[1]);                     \
        if (d_off[2] >= 0) dp2 = *reinterpret_cast<const float4*>(s_ + d_off[2]);                     \
        if (d_off[3] >= 0) dp3 = *reinterpret_cast<const float4*>(s_ + d_off[3]);                     \
    }
#define CTI_MBB_PUT_T()                                                                               \
    {                                                                                                 \
        if (t < t4n) *reinterpret_cast<float4*>(Rg + tput0) = tp0;                                    \
        if (t + nthr < t4n) *reinterpret_cast<float4*>(Rg + tput1) = tp1;                             \
    }
    CTI_MBB_PF_T(r_lo)
    CTI_MBB_PF_D(r_lo)
    for (int r = r_lo; r < r_hi; ++r) {
        __syncthreads();                                                        // previous rank's dVr / dT readers are done
        CTI_MBB_PUT_T()
        for (int e = t; e < V * HR; e += nthr) Vs[e] = vb[(int64_t)(e / HR) * K + r * HR + (e % HR)];
        for (int e = t; e < Q * HR; e += nthr) Qs[e] = qb[(int64_t)(e / HR) * K + r * HR + (e % HR)];
        __syncthreads();
        // X[v][g][j][k]: the thread's T_eff column in registers, v strided over the column groups
        if (xsp < xns && !(CTI_MBB_SKIP & 1)) {
            float tcol[HR];
#pragma unroll
            for (int i = 0; i < HR; ++i) tcol[i] = Rg[i * TP + xc];
            if ((inner & 63) == 0) {
                // a wave shares its v (inner % 64 == 0): Vr[v][0..HR) comes through the scalar path (s_load -> SGPR operands), no LDS traffic
                const int vfirst = __builtin_amdgcn_readfirstlane(xsp);
                const float* vsrc = vb + r * HR;
                for (int v = vfirst; v < V; v += xns) {
                    const float* vr = vsrc + (int64_t)v * K;
                    float x = 0.f;
#pragma unroll
                    for (int i = 0; i < HR; ++i) x = fmaf(tcol[i], vr[i], x);
                    Xs[v * G * HH + x_off] = x;
                }
            } else {
                for (int v = xsp; v < V; v += xns) {
                    float x = 0.f;
#pragma unroll
                    for (int i = 0; i < HR; ++i) x = fmaf(tcol[i], Vs[v * HR + i], x);
                    Xs[v * G * HH + x_off] = x;
                }
            }
        }
        __syncthreads();                                                        // T readers done: the region becomes the dM slice
        if (d_off[0] >= 0) *reinterpret_cast<float4*>(Rg + d_dst[0]) = dp0;
        if (d_off[1] >= 0) *reinterpret_cast<float4*>(Rg + d_dst[1]) = dp1;
        if (d_off[2] >= 0) *reinterpret_cast<float4*>(Rg + d_dst[2]) = dp2;
        if (d_off[3] >= 0) *reinterpret_cast<float4*>(Rg + d_dst[3]) = dp3;
        __syncthreads();
        if (r + 1 < r_hi) CTI_MBB_PF_D(r + 1)                                   // next rank's slice flies under this rank's arithmetic
        // dQr[q][j] = sum_{v,g} sum_k dM[v,q,g,k] X[v,g,j,k]: P lanes split v, shuffle-reduced
        if (t < nout * P && !(CTI_MBB_SKIP & 2)) {
            float s = 0.f;
            for (int v = qpart; v < V; v += P)
                for (int g = 0; g < G; ++g) {
                    const float* dm = Rg + ((v * Q + qq) * G + g) * HP;
                    const float* xr = Xs + (v * G + g) * HH + qj * HR;
#pragma unroll
                    for (int k4 = 0; k4 < HR; k4 += 4) {
                        const float4 a = *reinterpret_cast<const float4*>(dm + k4), x4 = *reinterpret_cast<const float4*>(xr + k4);
                        s = fmaf(a.x, x4.x, s); s = fmaf(a.y, x4.y, s); s = fmaf(a.z, x4.z, s); s = fmaf(a.w, x4.w, s);
                    }
                }
            for (int off = P >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (qpart == 0) dQr[((int64_t)b * Q + qq) * K + r * HR + qj] = s;
        }
        __syncthreads();
        // dX[v][g][j][k] = sum_q dM[v,q,g,k] Qr[q,j]  (overwrites X)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int k = dx_k[n], g = dx_g[n], v = dx_v[n];
            if (v >= 0 && !(CTI_MBB_SKIP & 4)) {
                float acc[HR];
#pragma unroll
                for (int j = 0; j < HR; ++j) acc[j] = 0.f;
                for (int q = 0; q < Q; ++q) {
                    const float m = Rg[((v * Q + q) * G + g) * HP + k];
#pragma unroll
                    for (int j4 = 0; j4 < HR; j4 += 4) {
                        const float4 q4 = *reinterpret_cast<const float4*>(Qs + q * HR + j4);
                        acc[j4] = fmaf(m, q4.x, acc[j4]); acc[j4 + 1] = fmaf(m, q4.y, acc[j4 + 1]);
                        acc[j4 + 2] = fmaf(m, q4.z, acc[j4 + 2]); acc[j4 + 3] = fmaf(m, q4.w, acc[j4 + 3]);
                    }
                }
#pragma unroll
                for (int j = 0; j < HR; ++j) Xs[(v * G + g) * HH + j * HR + k] = acc[j];
            }
        }
        __syncthreads();                                                        // slice readers done: the region becomes T again
        CTI_MBB_PUT_T()
        __syncthreads();
        if (r + 1 < r_hi) CTI_MBB_PF_T(r + 1)
        // dVr[v][i] = sum_{g,jk} dX[v][g][jk] T[i][jk*G + g].  Item = (v pair, i quad, 1 of 8 interleaved column slices): 8
        // accumulators per thread, 6 LDS reads per 8 FMAs, then a 3-step shuffle over the slice lanes (lane bits 0..2).
        if (!(CTI_MBB_SKIP & 8)) {
            const int slice = t & 7, iq = (t >> 3) & (HR / 4 - 1), vp = t / (8 * (HR / 4));
            if (vp < (V + 1) / 2) {
                const int v0 = 2 * vp, v1 = min(V - 1, v0 + 1), i0 = iq * 4;
                float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
                if (G == 2) {
                    // four consecutive jk and both g per trip: T[i][(jk..jk+3)*2 + g] are 8 consecutive floats (two 16-B reads per i), the dX rows
                    // of g = 0 / 1 one 16-B read each: 12 reads of 16 B feed 64 FMAs (the generic loop issues 48 reads of 4 B)
                    for (int jb = slice; jb < HH / 4; jb += 8) {
                        const int jk = jb * 4;
                        const float4 x00 = *reinterpret_cast<const float4*>(Xs + (v0 * 2 + 0) * HH + jk), x01 = *reinterpret_cast<const float4*>(Xs + (v0 * 2 + 1) * HH + jk);
                        const float4 x10 = *reinterpret_cast<const float4*>(Xs + (v1 * 2 + 0) * HH + jk), x11 = *reinterpret_cast<const float4*>(Xs + (v1 * 2 + 1) * HH + jk);
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            const float* tr = Rg + (i0 + n) * TP + jk * 2;
                            const float4 ta = *reinterpret_cast<const float4*>(tr), tb = *reinterpret_cast<const float4*>(tr + 4);
                            // ta = (jk,g0) (jk,g1) (jk+1,g0) (jk+1,g1);  tb = (jk+2,g0) (jk+2,g1) (jk+3,g0) (jk+3,g1)
                            a0[n] += x00.x * ta.x + x01.x * ta.y + x00.y * ta.z + x01.y * ta.w + x00.z * tb.x + x01.z * tb.y + x00.w * tb.z + x01.w * tb.w;
                            a1[n] += x10.x * ta.x + x11.x * ta.y + x10.y * ta.z + x11.y * ta.w + x10.z * tb.x + x11.z * tb.y + x10.w * tb.z + x11.w * tb.w;
                        }
                    }
                } else
                for (int e = slice; e < inner; e += 8) {
                    const int g = e / HH, jk = e - g * HH;
                    const float x0 = Xs[(v0 * G + g) * HH + jk], x1 = Xs[(v1 * G + g) * HH + jk];
                    const float* tr = Rg + i0 * TP + jk * G + g;
#pragma unroll
                    for (int n = 0; n < 4; ++n) { const float tv = tr[n * TP]; a0[n] = fmaf(x0, tv, a0[n]); a1[n] = fmaf(x1, tv, a1[n]); }
                }
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int off = 4; off > 0; off >>= 1) { a0[n] += __shfl_xor(a0[n], off, 64); a1[n] += __shfl_xor(a1[n], off, 64); }
                if (slice == 0) {
                    float* o0 = dVr + ((int64_t)b * V + v0) * K + r * HR + i0;
                    *reinterpret_cast<float4*>(o0) = make_float4(a0[0], a0[1], a0[2], a0[3]);
                    if (v0 + 1 < V) *reinterpret_cast<float4*>(o0 + K) = make_float4(a1[0], a1[1], a1[2], a1[3]);
                }
            }
        }
        // dT_b[r][i][c] = sum_v Vr[v][i] dX[v][c]: a thread owns column c and HR / tistep consecutive i: one X read feeds them all
        if (!(CTI_MBB_SKIP & 16) && ti0 < tistep) {
            const int ni = HR / tistep, i0 = ti0 * ni;                                   // tistep in {1, 2, 4, ...} divides HR
            float acc[HR];
#pragma unroll
            for (int n = 0; n < HR; ++n) acc[n] = 0.f;
            if (ni == 8) {                                                   // hr = 16, G = 2: two 16-B reads of the Vr row per object
                for (int v = 0; v < V; ++v) {
                    const float x = Xs[(v * G + tg) * HH + tjk];
                    const float4 va = *reinterpret_cast<const float4*>(Vs + v * HR + i0), vb4 = *reinterpret_cast<const float4*>(Vs + v * HR + i0 + 4);
                    acc[0] = fmaf(va.x, x, acc[0]); acc[1] = fmaf(va.y, x, acc[1]); acc[2] = fmaf(va.z, x, acc[2]); acc[3] = fmaf(va.w, x, acc[3]);
                    acc[4] = fmaf(vb4.x, x, acc[4]); acc[5] = fmaf(vb4.y, x, acc[5]); acc[6] = fmaf(vb4.z, x, acc[6]); acc[7] = fmaf(vb4.w, x, acc[7]);
                }
            } else {
                for (int v = 0; v < V; ++v) {
                    const float x = Xs[(v * G + tg) * HH + tjk];
                    const float* vr = Vs + v * HR + i0;
#pragma unroll
                    for (int n = 0; n < HR; ++n) if (n < ni) acc[n] = fmaf(vr[n], x, acc[n]);
                }
            }
            float* o = dTpart + ((int64_t)b * R + r) * HR * inner + (int64_t)i0 * inner + tc;
#pragma unroll
            for (int n = 0; n < HR; ++n) if (n < ni) o[(int64_t)n * inner] = acc[n];
        }
    }
#undef CTI_MBB_PF_T
#undef CTI_MBB_PF_D
#undef CTI_MBB_PUT_T
}

// =====================================================================================================================
// M build backward on the matrix cores (round 3; hr = 16, G = 2, V <= 48, Q <= 16: every model configuration).  The staged kernel above is fp32 VALU
// work -- 0.90 ms of the 5.6-ms training step at B = 256.  All five contractions of a rank are small GEMMs whose 16-wide axes fit the 16x16x16 bf16 MFMA
// (operands split into bf16 hi + lo in registers, three products per pair: fp32-grade, like the forward's M build):
//   1  X[v, c]  = sum_i Vr[v, i] T[i, c]                  3 v tiles x 32 c tiles (c = (g, j) x k), one K step
//   2  dQr[q, j] += sum_k dM[(v, q, g), k] X[v, g, j, k]   per (v, g): A = the dM rows of the pair, B = the X tile            } one pass, wave-local: the lane
//   3  dX[v, g, j, k] = sum_q dM[(v, q, g), k] Qr[q, j]    per (v, g): A = the same dM rows transposed, B = Qr; OVERWRITES X } that read X[..] writes dX[..] there
//   4  dVr[v, i] = sum_c dX[v, c] T[i, c]                  3 v tiles, K = 512 in 32 steps over 15 waves, partials summed through LDS
//   5  dT[i, c]  = sum_v Vr[v, i] dX[v, c]                 32 c tiles, K = V in 3 steps; written straight to the per-sample partial
// LDS: one region time-shared by T[r] (rows 516 floats apart, (g, j, k) order: de-interleaved from the (j, k, g) order of T_eff on the way in) and the
// rank's dM slice (rows 20 floats apart), as in the staged kernel; X / dX as [v][516] with the four 16-B chunks of a 16-float (g, j) row XOR-swizzled by
// (j >> 2) & 3 so that the 16 lanes of an operand read (stride 64 B) and the accumulator stores hit distinct banks.  T[r + 1] and the next dM slice are
// prefetched into registers (8 + 16 per thread) under the arithmetic of rank r.
typedef short bw_s16x4 __attribute__((ext_vector_type(4)));
typedef float bw_f32x4 __attribute__((ext_vector_type(4)));
template <int TERMS>
__device__ __forceinline__ void bw_split4(const bw_f32x4 x, bw_s16x4& hi, bw_s16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = static_cast<__bf16>(x[e]);
        hi[e] = __builtin_bit_cast(short, h);
        lo[e] = TERMS == 3 ? __builtin_bit_cast(short, static_cast<__bf16>(x[e] - static_cast<float>(h))) : (short)0;
    }
}
template <int TERMS>
__device__ __forceinline__ bw_f32x4 bw_mfma(const bw_s16x4 ah, const bw_s16x4 al, const bw_s16x4 bh, const bw_s16x4 bl, bw_f32x4 c) {
    if (TERMS == 3) {
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, c, 0, 0, 0);
}
constexpr int MBM_PV = 516;          // row pitch (floats) of Xs[v] and of T[i] in LDS: 4 * 516 = 16 (mod 64) banks between the four row groups of a lane quartet
constexpr int MBM_HP = 20;           // row pitch of the dM slice
constexpr int MBM_T_FLOATS = 16 * MBM_PV, MBM_SCR_Q = 16 * 256, MBM_SCR_V = 15 * 256;

// Operands that are read more than once live in LDS PRE-SPLIT: one dword per element = bf16 hi | bf16 lo << 16 (the same 4 B as the fp32 value).  A consumer
// turns four dwords into the MFMA's hi and lo operands with four v_perm_b32 instead of ~20 VALU per split -- the splits, not the MFMAs, were the kernel's bound
// (T is split once per workgroup, the dM slice once per sample instead of twice, dX once instead of twice, Vr once instead of twice).
template <int TERMS>
__device__ __forceinline__ float bw_pack(float x) {
    const __bf16 h = static_cast<__bf16>(x);
    const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, h);
    if (TERMS != 3) return __builtin_bit_cast(float, hb);
    const unsigned lb = (unsigned)__builtin_bit_cast(unsigned short, static_cast<__bf16>(x - static_cast<float>(h)));
    return __builtin_bit_cast(float, hb | (lb << 16));
}
template <int TERMS>
__device__ __forceinline__ bw_f32x4 bw_pack4(const bw_f32x4 x) {
    const float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    const bw_f32x4 o = {bw_pack<TERMS>(x0), bw_pack<TERMS>(x1), bw_pack<TERMS>(x2), bw_pack<TERMS>(x3)};
    return o;
}
template <int TERMS>
__device__ __forceinline__ void bw_unpack4(const bw_f32x4 p, bw_s16x4& hi, bw_s16x4& lo) {
    // (through scalar copies: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the index -- hipcc 7.2, found the hard way)
    const float f0 = p[0], f1 = p[1], f2 = p[2], f3 = p[3];
    const unsigned u0 = __builtin_bit_cast(unsigned, f0), u1 = __builtin_bit_cast(unsigned, f1), u2 = __builtin_bit_cast(unsigned, f2), u3 = __builtin_bit_cast(unsigned, f3);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 h = {__builtin_amdgcn_perm(u1, u0, 0x05040100u), __builtin_amdgcn_perm(u3, u2, 0x05040100u)};
    hi = __builtin_bit_cast(bw_s16x4, h);
    if (TERMS == 3) {
        const u32x2 l = {__builtin_amdgcn_perm(u1, u0, 0x07060302u), __builtin_amdgcn_perm(u3, u2, 0x07060302u)};
        lo = __builtin_bit_cast(bw_s16x4, l);
    } else lo = hi;
}

template <int TERMS>
__global__ __launch_bounds__(1024) void mbuild_bwd_mfma_kernel(const float* __restrict__ dM, const float* __restrict__ Vr, const float* __restrict__ Qr,
                                                               const float* __restrict__ Teff, float* __restrict__ dVr, float* __restrict__ dQr,
                                                               float* __restrict__ dTpart, int B, int V, int Q, int R, int bpc, int region_floats) {
    constexpr int HR = 16, G = 2, inner = HR * HR * G, PV = MBM_PV, HP = MBM_HP;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* R1 = sm;                              // T[r] ([16][PV]) + reduction scratch, or the dM slice ([V*Q*G][HP])
    float* Xs = R1 + region_floats;              // [V][PV]: X, later dX; a (g, j) row of 16 k values at (g*16 + j)*16, chunks swizzled
    float* Vs = Xs + (size_t)V * PV;             // [48][16], rows >= V zero
    float* Qs = Vs + 48 * HR;                    // [16][16], rows >= Q zero
    float* scrQ = R1 + MBM_T_FLOATS;             // [16 waves][16 q][16 j]   partial dQr tiles
    float* scrV = scrQ + MBM_SCR_Q;              // [15 waves][16 v][16 i]   partial dVr tiles
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, l15 = lane & 15, l4 = lane >> 4;
    // workgroup = (rank r, chunk of bpc samples): T[r] is loaded ONCE (8 registers per thread, re-put into the time-shared region twice per sample) and the
    // dT tiles accumulate over the chunk in registers -- one partial per (chunk, rank) instead of one per (sample, rank): 8 MB instead of 268 MB at B = 256
    const int r = blockIdx.x;
    const int b_lo = blockIdx.y * bpc, b_hi = min(B, b_lo + bpc);
    const int K = R * HR;
    const int rowsM = V * Q * G;
    const bw_f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // prefetch registers: T[r] = 2 048 16-B pieces (2 per thread), the dM slice = rowsM * 4 pieces (<= 4 per thread: rowsM <= 1 024); piece f = t + 1024 u is
    // floats part*4 .. +3 of row f >> 2 (addresses recomputed where they are used: index arrays would cost 12 registers of the 128)
    const int nD4 = rowsM * 4;
    bw_f32x4 tp[2] = {z4, z4}, dp[4] = {z4, z4, z4, z4};
    {
        const bw_f32x4* Tr = reinterpret_cast<const bw_f32x4*>(Teff + (int64_t)r * HR * inner);
        tp[0] = bw_pack4<TERMS>(Tr[(unsigned)t]); tp[1] = bw_pack4<TERMS>(Tr[(unsigned)t + 1024u]);       // split ONCE per workgroup
    }
    auto pf_D = [&](int bb) {
        const float* s_ = dM + (int64_t)bb * rowsM * K + r * HR;
        int tz = t;
        asm volatile("" : "+v"(tz));            // lane offsets recomputed per call: hoisted out of the rank loop they spill, and a scratch reload waits for vmcnt(0),
                                                // i.e. serialises the four prefetch loads it sits between
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int f = tz + u * 1024; const unsigned off = (unsigned)((f >> 2) * K + (f & 3) * 4); if (f < nD4) dp[u] = *reinterpret_cast<const bw_f32x4*>(s_ + off); }
    };
    auto put_T = [&]() {                         // piece f: i = f / 128, c0 = (f % 128) * 4 = (jk0, g0) (jk0, g1) (jk0 + 1, g0) (jk0 + 1, g1) -> [i][g * 256 + jk]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = t + u * 1024, i = f >> 7, jk0 = (f & 127) * 2;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 g0 = {tp[u][0], tp[u][2]}, g1 = {tp[u][1], tp[u][3]};
            *reinterpret_cast<f32x2*>(R1 + i * PV + jk0) = g0;
            *reinterpret_cast<f32x2*>(R1 + i * PV + 256 + jk0) = g1;
        }
    };
    // this thread's element of the sample's Vr / Qr slice (rows beyond V / Q: zero), loaded one sample ahead like the dM slice: read at the top of the
    // iteration it cost an exposed global-load latency per sample
    float vq_pf = 0.f;
    auto pf_VQ = [&](int bb) {
        int tz = t;
        asm volatile("" : "+v"(tz));
        vq_pf = 0.f;
        if (tz < 48 * HR) { const int v = tz >> 4, i = tz & 15; if (v < V) vq_pf = (Vr + (int64_t)bb * V * K + r * HR)[(unsigned)(v * K + i)]; }
        else { const int e = tz - 48 * HR, q = e >> 4, j = e & 15; if (q < Q) vq_pf = (Qr + (int64_t)bb * Q * K + r * HR)[(unsigned)(q * K + j)]; }
    };
    pf_D(b_lo);
    pf_VQ(b_lo);
    bw_f32x4 accT[2] = {z4, z4};
    for (int b = b_lo; b < b_hi; ++b) {
        // the lane coordinates are re-derived from an opaque copy in every iteration: as loop invariants the ~25 LDS / global lane offsets of the five phases
        // are all hoisted, and what does not fit the 128 registers is spilled (a reload waits for vmcnt(0): it stalls on the prefetch in flight)
        int ln_ = lane;
        asm volatile("" : "+v"(ln_));
        const int l15 = ln_ & 15, l4 = ln_ >> 4;
        __syncthreads();                                                        // previous sample: phase 5 / the reductions are done with T, Vs, Qs, scratch
        put_T();
        // (wave-uniform 64-bit bases + 32-bit lane offsets everywhere: per-lane 64-bit addresses hoisted out of the rank loop were what spilled, and a
        // scratch reload waits for vmcnt(0) -- i.e. for the prefetches in flight)
        Vs[t] = bw_pack<TERMS>(vq_pf);                                           // [48][16] then [16][16]: Qs = Vs + 768 (the element was loaded one sample ahead)
        __syncthreads();
        if (b + 1 < b_hi) pf_VQ(b + 1);
        // ---- 1: X = Vr T.  Wave w: c tiles 2w, 2w + 1 against the three v tiles
        {
            bw_s16x4 ah[3], al[3];
#pragma unroll
            for (int vt = 0; vt < 3; ++vt) bw_unpack4<TERMS>(*reinterpret_cast<const bw_f32x4*>(Vs + (vt * 16 + l15) * HR + 4 * l4), ah[vt], al[vt]);
#pragma unroll 1
            for (int n = 0; n < 2; ++n) {
                const int ct = 2 * wid + n;
                bw_f32x4 tb;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) tb[ii] = R1[(4 * l4 + ii) * PV + ct * 16 + l15];
                bw_s16x4 bh, bl;
                bw_unpack4<TERMS>(tb, bh, bl);
                const int sw = ct * 16 + ((((l15 >> 2) ^ ((ct >> 2) & 3)) << 2) | (l15 & 3));     // (j >> 2) & 3 == (ct >> 2) & 3
#pragma unroll
                for (int vt = 0; vt < 3; ++vt) {
                    const bw_f32x4 x = bw_mfma<TERMS>(ah[vt], al[vt], bh, bl, z4);      // rows v = 16 vt + 4 l4 + ii, column k = l15 of tile ct
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) { const int v = vt * 16 + 4 * l4 + ii; if (v < V) Xs[v * PV + sw] = x[ii]; }
                }
            }
        }
        __syncthreads();                                                        // T readers done: the region becomes the dM slice
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int f = t + u * 1024; if (f < nD4) *reinterpret_cast<bw_f32x4*>(R1 + (f >> 2) * HP + (f & 3) * 4) = bw_pack4<TERMS>(dp[u]); }
        __syncthreads();
        // ---- 2 + 3: per (v, g) pair, wave-local.  dQr tile (rows q, columns j) accumulates over the wave's pairs; dX overwrites X in place
        bw_f32x4 accQ = z4;
        {
            bw_f32x4 qv;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) qv[ii] = Qs[(4 * l4 + ii) * HR + l15];       // B of phase 3: column j = l15, K = q (rows >= Q are zero)
            bw_s16x4 qh, ql;
            bw_unpack4<TERMS>(qv, qh, ql);
            const int qrow = min(l15, Q - 1);                                        // rows q >= Q of the dQr tile are never stored
            for (int p = wid; p < V * G; p += 16) {
                const int v = p >> 1, g = p & 1;
                float* xt = Xs + v * PV + (g * 16 + l15) * 16 + ((l4 ^ ((l15 >> 2) & 3)) << 2);     // X[v, g, j = l15, k = 4 l4 ..]: read here, dX written here
                bw_s16x4 a2h, a2l, b2h, b2l, a3h, a3l;
                bw_unpack4<TERMS>(*reinterpret_cast<const bw_f32x4*>(R1 + ((v * Q + qrow) * G + g) * HP + 4 * l4), a2h, a2l);
                bw_split4<TERMS>(*reinterpret_cast<const bw_f32x4*>(xt), b2h, b2l);
                accQ = bw_mfma<TERMS>(a2h, a2l, b2h, b2l, accQ);
                bw_f32x4 a3;                                                           // A of phase 3: row k = l15, K = q = 4 l4 + ii (clamped: Qr's zero rows cancel it)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) a3[ii] = R1[((v * Q + min(4 * l4 + ii, Q - 1)) * G + g) * HP + l15];
                bw_unpack4<TERMS>(a3, a3h, a3l);
                const bw_f32x4 d3 = bw_mfma<TERMS>(a3h, a3l, qh, ql, z4);           // rows k = 4 l4 + ii, column j = l15: this lane's four k are ONE 16-B chunk of row (g, j)
                *reinterpret_cast<bw_f32x4*>(xt) = bw_pack4<TERMS>(d3);              // dX is stored pre-split: phases 4 and 5 both read it
            }
        }
        __syncthreads();                                                        // slice readers done: the region becomes T again (+ scratch behind it)
        put_T();
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) scrQ[wid * 256 + (4 * l4 + ii) * 16 + l15] = accQ[ii];
        __syncthreads();
        if (b + 1 < b_hi) pf_D(b + 1);                                          // the next sample's dM slice flies under phases 4 + 5 (issued here, not earlier: 16 registers)
        // ---- 4: dVr = dX T^T.  Waves 0..14: v tile w / 5, K steps (w % 5), + 5, ... of the 32
        if (wid < 15) {
            const int vt = wid / 5;
            const int vrow = min(vt * 16 + l15, V - 1);
            bw_f32x4 accV = z4;
            for (int ks = wid - vt * 5; ks < 32; ks += 5) {
                bw_s16x4 a4h, a4l, b4h, b4l;
                bw_unpack4<TERMS>(*reinterpret_cast<const bw_f32x4*>(Xs + vrow * PV + ks * 16 + ((l4 ^ ((ks >> 2) & 3)) << 2)), a4h, a4l);
                bw_unpack4<TERMS>(*reinterpret_cast<const bw_f32x4*>(R1 + l15 * PV + ks * 16 + 4 * l4), b4h, b4l);
                accV = bw_mfma<TERMS>(a4h, a4l, b4h, b4l, accV);
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) scrV[wid * 256 + (4 * l4 + ii) * 16 + l15] = accV[ii];
        }
        // ---- 5: dT = Vr^T dX.  Wave w: c tiles 2w, 2w + 1; K = v in three steps
        {
            bw_s16x4 a5h[3], a5l[3];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                bw_f32x4 av;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) av[ii] = Vs[(s3 * 16 + 4 * l4 + ii) * HR + l15];      // row i = l15, K = v (rows >= V are zero)
                bw_unpack4<TERMS>(av, a5h[s3], a5l[s3]);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int ct = 2 * wid + n;
                const int sw = ct * 16 + ((((l15 >> 2) ^ ((ct >> 2) & 3)) << 2) | (l15 & 3));
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    bw_f32x4 xv;
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) xv[ii] = Xs[min(s3 * 16 + 4 * l4 + ii, V - 1) * PV + sw];
                    bw_s16x4 b5h, b5l;
                    bw_unpack4<TERMS>(xv, b5h, b5l);
                    accT[n] = bw_mfma<TERMS>(a5h[s3], a5l[s3], b5h, b5l, accT[n]);          // accumulates over the chunk's samples
                }
            }
        }
        __syncthreads();
        // ---- the partial tiles of phases 2 and 4
        int tq = t;
        asm volatile("" : "+v"(tq));                                            // (as in pf_D: keeps the store addresses out of the loop-invariant set)
        if (t < 256) {
            float s_ = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) s_ += scrQ[w * 256 + t];
            const int q = tq >> 4, j = tq & 15;
            if (q < Q) (dQr + (int64_t)b * Q * K + r * HR)[(unsigned)(q * K + j)] = s_;
        } else {
            const int e = tq - 256, vt = e >> 8, idx = e & 255, v = vt * 16 + (idx >> 4), i = idx & 15;
            float s_ = 0.f;
#pragma unroll
            for (int w = 0; w < 5; ++w) s_ += scrV[(vt * 5 + w) * 256 + idx];
            if (v < V) (dVr + (int64_t)b * V * K + r * HR)[(unsigned)(v * K + i)] = s_;
        }
    }
    float* o = dTpart + ((int64_t)blockIdx.y * R + r) * HR * inner;             // this chunk's partial of dT_eff[r]: rows i, column (j, k = l15, g) in T_eff's order
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int ct = 2 * wid + n, g = ct >> 4, j = ct & 15;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) o[(unsigned)((4 * l4 + ii) * inner + (j * 16 + l15) * G + g)] = accT[n][ii];
    }
}

// =====================================================================================================================
// softmax backward: dl = p * (dp - sum p*dp) over the softmax axis.
// =====================================================================================================================
constexpr int GMAXB = 8;
__global__ __launch_bounds__(256) void tri_sm_bwd_partial(const float* __restrict__ p, const float* __restrict__ dp, float* __restrict__ part,
                                                          int64_t N, int G, int64_t chunk_n, int nchunk) {
    __shared__ float red[4][GMAXB];
    const int b = blockIdx.y, c = blockIdx.x, t = threadIdx.x;
    const int64_t lo = (int64_t)c * chunk_n, hi = min(N, lo + chunk_n);
    const float* pp = p + (int64_t)b * N * G;
    const float* dd = dp + (int64_t)b * N * G;
    for (int g0 = 0; g0 < G; g0 += GMAXB) {
        float s[GMAXB];
#pragma unroll
        for (int g = 0; g < GMAXB; ++g) s[g] = 0.f;
        for (int64_t n = lo + t; n < hi; n += 256)
#pragma unroll
            for (int g = 0; g < GMAXB; ++g)
                if (g0 + g < G) s[g] = fmaf(pp[n * G + g0 + g], dd[n * G + g0 + g], s[g]);
#pragma unroll
        for (int g = 0; g < GMAXB; ++g) s[g] = wave_sum(s[g]);
        if ((t & 63) == 0)
#pragma unroll
            for (int g = 0; g < GMAXB; ++g) red[t >> 6][g] = s[g];
        __syncthreads();
        if (t < GMAXB && g0 + t < G) part[((int64_t)b * nchunk + c) * G + g0 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void tri_sm_bwd_apply(const float* __restrict__ p, const float* __restrict__ dp, const float* __restrict__ part,
                                                        float* __restrict__ dl, int64_t NG, int G, int nchunk) {
    __shared__ float s[64];                                  // G <= 64 per pass
    const int b = blockIdx.y;
    for (int g = threadIdx.x; g < G && g < 64; g += 256) {
        float a = 0.f;
        for (int c = 0; c < nchunk; ++c) a += part[((int64_t)b * nchunk + c) * G + g];
        s[g] = a;
    }
    __syncthreads();
    const float* pp = p + (int64_t)b * NG;
    const float* dd = dp + (int64_t)b * NG;
    float* o = dl + (int64_t)b * NG;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < NG; f += stride) {
        const int g = (int)(f % G);
        o[f] = pp[f] * (dd[f] - s[g]);
    }
}
__global__ __launch_bounds__(256) void bi_sm_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, float* __restrict__ dl,
                                                        int rows, int N) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pp = p + (int64_t)row * N;
    const float* dd = dp + (int64_t)row * N;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s = fmaf(pp[n], dd[n], s);
    s = wave_sum(s);
    for (int n = lane; n < N; n += 64) dl[(int64_t)row * N + n] = pp[n] * (dd[n] - s);
}

// =====================================================================================================================
// tri pool backward.  out[b,d] = sum_v vt[v,d] sum_q qt[q,d] sum_a w[v,q,a] at[a,d]
// kernel 1 (lane = d): dvt, dqt, dat;  kernel 2 (workgroup per (b,v)): dw[v,q,a] = sum_d do vt qt at
// =====================================================================================================================
__global__ __launch_bounds__(256) void tri_pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ vt, const float* __restrict__ qt,
                                                           const float* __restrict__ at, const float* __restrict__ w, int64_t w_sb, int64_t w_sv,
                                                           int64_t w_sq, int64_t w_sa, float* __restrict__ dvt, float* __restrict__ dqt,
                                                           float* __restrict__ dat, int V, int Q, int A, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y, t = threadIdx.x, d = blockIdx.x * 256 + t;
    const bool live = d < D;
    const int dd = live ? d : D - 1;
    float* qs = sm; float* as = qs + Q * 256; float* dq = as + A * 256; float* da = dq + Q * 256;
    const float* qb = qt + (int64_t)b * Q * D; const float* ab = at + (int64_t)b * A * D; const float* vb = vt + (int64_t)b * V * D;
    for (int q = 0; q < Q; ++q) { qs[q * 256 + t] = qb[(int64_t)q * D + dd]; dq[q * 256 + t] = 0.f; }
    for (int a = 0; a < A; ++a) { as[a * 256 + t] = ab[(int64_t)a * D + dd]; da[a * 256 + t] = 0.f; }
    const float g = dout[(int64_t)b * D + dd];
    const float* wb = w + (int64_t)b * w_sb;
    for (int v = 0; v < V; ++v) {
        const float x = vb[(int64_t)v * D + dd], gx = g * x;
        float sv = 0.f;
        for (int q = 0; q < Q; ++q) {
            const float* wr = wb + v * w_sv + q * w_sq;
            const float qq = qs[q * 256 + t];
            float tq = 0.f;
            for (int a = 0; a < A; ++a) {
                const float ww = wr[a * w_sa];
                tq = fmaf(ww, as[a * 256 + t], tq);
                da[a * 256 + t] = fmaf(gx * qq, ww, da[a * 256 + t]);
            }
            sv = fmaf(qq, tq, sv);
            dq[q * 256 + t] = fmaf(gx, tq, dq[q * 256 + t]);
        }
        if (live) dvt[((int64_t)b * V + v) * D + d] = g * sv;
    }
    if (live) {
        for (int q = 0; q < Q; ++q) dqt[((int64_t)b * Q + q) * D + d] = dq[q * 256 + t];
        for (int a = 0; a < A; ++a) dat[((int64_t)b * A + a) * D + d] = da[a * 256 + t];
    }
}
// Small-shape form of the kernel above (A <= 8, Q <= 16: the model configurations): the attention slice is compacted into LDS once
// ([v][q][AP], one broadcast 16-B read per (v,q)), qt / at columns and the dqt / dat accumulators live in registers -- the generic
// kernel reads w through dependent scalar loads and keeps its accumulators in LDS (217 us at B = 256; this one ~1/4 of that).
template <int AP>
__global__ __launch_bounds__(256) void tri_pool_bwd_small_kernel(const float* __restrict__ dout, const float* __restrict__ vt, const float* __restrict__ qt,
                                                                 const float* __restrict__ at, const float* __restrict__ w, int64_t w_sb, int64_t w_sv,
                                                                 int64_t w_sq, int64_t w_sa, float* __restrict__ dvt, float* __restrict__ dqt,
                                                                 float* __restrict__ dat, int V, int Q, int A, int D) {
    constexpr int QM = 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [V*Q][AP]
    const int b = blockIdx.y, t = threadIdx.x, d = blockIdx.x * 256 + t;
    const bool live = d < D;
    const int dd = live ? d : D - 1;
    const float* wb = w + (int64_t)b * w_sb;
    for (int i = t; i < V * Q * AP; i += 256) {
        const int a = i % AP, vq = i / AP, q = vq % Q, v = vq / Q;
        sm[i] = a < A ? wb[v * w_sv + q * w_sq + a * w_sa] : 0.f;
    }
    float qr[QM], dq[QM], ar[AP], da[AP];
#pragma unroll
    for (int q = 0; q < QM; ++q) { qr[q] = q < Q ? qt[((int64_t)b * Q + q) * D + dd] : 0.f; dq[q] = 0.f; }
#pragma unroll
    for (int a = 0; a < AP; ++a) { ar[a] = a < A ? at[((int64_t)b * A + a) * D + dd] : 0.f; da[a] = 0.f; }
    const float g = dout[(int64_t)b * D + dd];
    __syncthreads();
    const float* vb = vt + (int64_t)b * V * D + dd;
    for (int v = 0; v < V; ++v) {
        const float gx = g * vb[(int64_t)v * D];
        const float4* wr = reinterpret_cast<const float4*>(sm + (size_t)v * Q * AP);
        float sv = 0.f;
#pragma unroll
        for (int q = 0; q < QM; ++q) {
            if (q < Q) {
                const float4 w0 = wr[q * (AP / 4)];
                const float gq = gx * qr[q];
                float tq = w0.x * ar[0] + w0.y * ar[1] + w0.z * ar[2] + w0.w * ar[3];
                da[0] = fmaf(gq, w0.x, da[0]); da[1] = fmaf(gq, w0.y, da[1]); da[2] = fmaf(gq, w0.z, da[2]); da[3] = fmaf(gq, w0.w, da[3]);
                if (AP == 8) {
                    const float4 w1 = wr[q * 2 + 1];
                    tq += w1.x * ar[AP - 4] + w1.y * ar[AP - 3] + w1.z * ar[AP - 2] + w1.w * ar[AP - 1];
                    da[AP - 4] = fmaf(gq, w1.x, da[AP - 4]); da[AP - 3] = fmaf(gq, w1.y, da[AP - 3]);
                    da[AP - 2] = fmaf(gq, w1.z, da[AP - 2]); da[AP - 1] = fmaf(gq, w1.w, da[AP - 1]);
                }
                sv = fmaf(qr[q], tq, sv);
                dq[q] = fmaf(gx, tq, dq[q]);
            }
        }
        if (live) dvt[((int64_t)b * V + v) * D + d] = g * sv;
    }
    if (live) {
#pragma unroll
        for (int q = 0; q < QM; ++q) if (q < Q) dqt[((int64_t)b * Q + q) * D + d] = dq[q];
#pragma unroll
        for (int a = 0; a < AP; ++a) if (a < A) dat[((int64_t)b * A + a) * D + d] = da[a];
    }
}
// dw[b,v,q,a] = sum_d x[d] * qt[b,q,d] * at[b,a,d], x[d] = dout[b,d/kdiv] * vt[b,v,d]; also serves the bi pool (A = 1, at == NULL)
// (PQC, PAC) = the per-lane block of (q, a) accumulators: (4, 8) for the tri pool, (16, 1) for the bi pool -- one sweep over D covers
// every q there, and no FMA is spent on padded answers
template <int PQC, int PAC>
__global__ __launch_bounds__(256) void pool_dw_kernel(const float* __restrict__ dout, const float* __restrict__ vt, const float* __restrict__ qt,
                                                      const float* __restrict__ at, float* __restrict__ dw, int V, int Q, int A, int D, int kdiv) {
    __shared__ float red[4][PQC * PAC];
    const int bv = blockIdx.x, b = bv / V, t = threadIdx.x;
    const float* vrow = vt + (int64_t)bv * D;
    const float* qb = qt + (int64_t)b * Q * D;
    const float* ab = at ? at + (int64_t)b * A * D : nullptr;
    const float* go = dout + (int64_t)b * (D / kdiv);
    for (int q0 = 0; q0 < Q; q0 += PQC)
        for (int a0 = 0; a0 < A; a0 += PAC) {
            float acc[PQC][PAC];
#pragma unroll
            for (int i = 0; i < PQC; ++i)
#pragma unroll
                for (int j = 0; j < PAC; ++j) acc[i][j] = 0.f;
            for (int d = t; d < D; d += 256) {
                const float x = go[d / kdiv] * vrow[d];
                float av[PAC];
#pragma unroll
                for (int j = 0; j < PAC; ++j) av[j] = ab ? ((a0 + j < A) ? ab[(int64_t)(a0 + j) * D + d] : 0.f) : (j == 0 ? 1.f : 0.f);
#pragma unroll
                for (int i = 0; i < PQC; ++i) {
                    const float xq = (q0 + i < Q) ? x * qb[(int64_t)(q0 + i) * D + d] : 0.f;
#pragma unroll
                    for (int j = 0; j < PAC; ++j) acc[i][j] = fmaf(xq, av[j], acc[i][j]);
                }
            }
#pragma unroll
            for (int i = 0; i < PQC; ++i)
#pragma unroll
                for (int j = 0; j < PAC; ++j) {
                    const float s = wave_sum(acc[i][j]);
                    if ((t & 63) == 0) red[t >> 6][i * PAC + j] = s;
                }
            __syncthreads();
            if (t < PQC * PAC) {
                const int i = t / PAC, j = t % PAC;
                if (q0 + i < Q && a0 + j < A) dw[((int64_t)bv * Q + q0 + i) * A + a0 + j] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
            }
            __syncthreads();
        }
}

// bi pool backward (lane = d): out[b,n] = sum_{j<k} sum_v vt[v,nk+j] sum_q w[v,q] qt[q,nk+j]
__global__ __launch_bounds__(256) void bi_pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ vt, const float* __restrict__ qt,
                                                          const float* __restrict__ w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                                                          float* __restrict__ dvt, float* __restrict__ dqt, int V, int Q, int D, int k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y, t = threadIdx.x, d = blockIdx.x * 256 + t;
    const bool live = d < D;
    const int dd = live ? d : D - 1;
    float* qs = sm; float* dq = qs + Q * 256;
    const float* qb = qt + (int64_t)b * Q * D; const float* vb = vt + (int64_t)b * V * D;
    for (int q = 0; q < Q; ++q) { qs[q * 256 + t] = qb[(int64_t)q * D + dd]; dq[q * 256 + t] = 0.f; }
    const int NO = D / k;
    const float g = (dd / k < NO) ? dout[(int64_t)b * NO + dd / k] : 0.f;       // channels of a ragged tail get no gradient
    if (w) {
        const float* wb = w + (int64_t)b * w_sb;
        for (int v = 0; v < V; ++v) {
            const float gx = g * vb[(int64_t)v * D + dd];
            float sv = 0.f;
            for (int q = 0; q < Q; ++q) {
                const float ww = wb[v * w_sv + q * w_sq];
                sv = fmaf(ww, qs[q * 256 + t], sv);
                dq[q * 256 + t] = fmaf(gx, ww, dq[q * 256 + t]);
            }
            if (live) dvt[((int64_t)b * V + v) * D + d] = g * sv;
        }
    } else {                                                   // w == 1
        float sq = 0.f, sv = 0.f;
        for (int q = 0; q < Q; ++q) sq += qs[q * 256 + t];
        for (int v = 0; v < V; ++v) sv += vb[(int64_t)v * D + dd];
        for (int v = 0; v < V; ++v) if (live) dvt[((int64_t)b * V + v) * D + d] = g * sq;
        for (int q = 0; q < Q; ++q) dq[q * 256 + t] = g * sv;
    }
    if (live) for (int q = 0; q < Q; ++q) dqt[((int64_t)b * Q + q) * D + d] = dq[q * 256 + t];
}

// The model's case (k = 1, weights given, Q <= 16): a thread owns two channels and keeps qt[q] and its dqt[q] accumulators in REGISTERS; the
// attention slice w[b] sits in LDS and is read as broadcasts.  The kernel above holds those per-thread vectors in LDS (three LDS operations and
// one global load of w per (v, q)): 56 us per call at B = 256, D = 1024 against ~25 us of HBM time.
__global__ __launch_bounds__(128) void bi_pool_bwd_reg_kernel(const float* __restrict__ dout, const float* __restrict__ vt, const float* __restrict__ qt,
                                                              const float* __restrict__ w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                                                              float* __restrict__ dvt, float* __restrict__ dqt, int V, int Q, int D) {
    constexpr int QX = 16;
    extern __shared__ __attribute__((aligned(16))) float ws[];       // [V][16], columns Q..15 zero
    const int b = blockIdx.y, t = threadIdx.x;
    const int d = (blockIdx.x * 128 + t) * 2;
    const bool live = d < D;
    const float* wb = w + (int64_t)b * w_sb;
    for (int i = t; i < V * QX; i += 128) { const int v = i >> 4, q = i & 15; ws[i] = q < Q ? wb[v * w_sv + q * w_sq] : 0.f; }
    float2 qs[QX], dq[QX];
#pragma unroll
    for (int q = 0; q < QX; ++q) {
        qs[q] = (live && q < Q) ? *reinterpret_cast<const float2*>(qt + ((int64_t)b * Q + q) * D + d) : make_float2(0.f, 0.f);
        dq[q] = make_float2(0.f, 0.f);
    }
    const float2 g = live ? *reinterpret_cast<const float2*>(dout + (int64_t)b * D + d) : make_float2(0.f, 0.f);
    __syncthreads();
    for (int v = 0; v < V; ++v) {
        const float2 x = live ? *reinterpret_cast<const float2*>(vt + ((int64_t)b * V + v) * D + d) : make_float2(0.f, 0.f);
        const float gx0 = g.x * x.x, gx1 = g.y * x.y;
        float s0 = 0.f, s1 = 0.f;
        const float4* wr = reinterpret_cast<const float4*>(ws + v * QX);
#pragma unroll
        for (int q4 = 0; q4 < QX / 4; ++q4) {
            const float4 ww = wr[q4];
            const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = q4 * 4 + j;
                s0 = fmaf(wv[j], qs[q].x, s0); s1 = fmaf(wv[j], qs[q].y, s1);
                dq[q].x = fmaf(gx0, wv[j], dq[q].x); dq[q].y = fmaf(gx1, wv[j], dq[q].y);
            }
        }
        if (live) *reinterpret_cast<float2*>(dvt + ((int64_t)b * V + v) * D + d) = make_float2(g.x * s0, g.y * s1);
    }
#pragma unroll
    for (int q = 0; q < QX; ++q)
        if (live && q < Q) *reinterpret_cast<float2*>(dqt + ((int64_t)b * Q + q) * D + d) = dq[q];
}

// =====================================================================================================================
// bilinear logits backward (lane = d).  logits[b,g,v,q] = hs * sum_d vt[v,d] h[g,d] qt[q,d] + hb[g]
//   dvt[v,d] = hs sum_{g,q} dl h[g,d] qt[q,d];  dqt[q,d] = hs sum_{g,v} dl h[g,d] vt[v,d];
//   dh_b[g,d] = hs sum_{v,q} dl vt[v,d] qt[q,d]   (per-sample partial of s * dL/dh_eff; summed over b afterwards)
// =====================================================================================================================
__global__ __launch_bounds__(256) void bi_logits_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ vt, const float* __restrict__ qt,
                                                            const float* __restrict__ h, const float* __restrict__ h_scale,
                                                            float* __restrict__ dvt, float* __restrict__ dqt, float* __restrict__ dhpart,
                                                            int G, int V, int Q, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y, t = threadIdx.x, d = blockIdx.x * 256 + t;
    const bool live = d < D;
    const int dd = live ? d : D - 1;
    float* vs = sm; float* qs = vs + V * 256; float* dv = qs + Q * 256; float* dq = dv + V * 256;
    const float* vb = vt + (int64_t)b * V * D; const float* qb = qt + (int64_t)b * Q * D;
    for (int v = 0; v < V; ++v) { vs[v * 256 + t] = vb[(int64_t)v * D + dd]; dv[v * 256 + t] = 0.f; }
    for (int q = 0; q < Q; ++q) { qs[q * 256 + t] = qb[(int64_t)q * D + dd]; dq[q * 256 + t] = 0.f; }
    const float hs = h_scale ? h_scale[0] : 1.f;
    const float* dlb = dl + (int64_t)b * G * V * Q;
    for (int g = 0; g < G; ++g) {
        const float hh = h[(int64_t)g * D + dd] * hs;
        float dh = 0.f;
        for (int v = 0; v < V; ++v) {
            const float x = vs[v * 256 + t];
            float sv = 0.f;
            for (int q = 0; q < Q; ++q) {
                const float gl = dlb[((int64_t)g * V + v) * Q + q];
                const float qq = qs[q * 256 + t];
                sv = fmaf(gl, qq, sv);
                dq[q * 256 + t] = fmaf(gl * hh, x, dq[q * 256 + t]);
            }
            dv[v * 256 + t] = fmaf(sv, hh, dv[v * 256 + t]);
            dh = fmaf(sv, x, dh);
        }
        if (live) dhpart[((int64_t)b * G + g) * D + d] = dh * hs;
    }
    if (live) {
        for (int v = 0; v < V; ++v) dvt[((int64_t)b * V + v) * D + d] = dv[v * 256 + t];
        for (int q = 0; q < Q; ++q) dqt[((int64_t)b * Q + q) * D + d] = dq[q * 256 + t];
    }
}

// out[r] = sum_c x[r, c]  (one wave per row)
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += x[row * cols + c];
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

inline bool aligned16b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <class K>
int set_lds(K kern, size_t bytes, const char* what) {
    if (bytes <= 64 * 1024) return CTI_OK;
    if (bytes > 160 * 1024) return fail(CTI_E_SHAPE, "%s: needs %zu B of LDS (> 160 KiB)", what, bytes);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e == hipSuccess ? CTI_OK : fail((int)e, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
}


// =====================================================================================================
// Backward of mode 3 + rank sum (out[b,v,q,a,g] = sum_k M[b,v,q,g,k] Ar[b,a,k], src/Tensor.py:16-20 with src/tc.py:50):
//   dM[b,v,q,g,k] = sum_a dout[b,v,q,a,g] Ar[b,a,k]        contraction over A (3 or 6): an element-wise pass that WRITES dM once
//   dAr[b,a,k]    = sum_{v,q,g} dout[b,v,q,a,g] M[b,v,q,g,k]  A output rows per sample: a streaming reduction that READS M once
// As GEMMs both need transposed copies of M / dout and plane splits of a (B, V*Q*G, K) tensor (528 MB at B = 256) for a contraction of
// length 3 resp. an output of 3 rows; here each is one pass at the HBM rate, exact fp32.
// =====================================================================================================
constexpr int CB_AMAX = 8;

__global__ __launch_bounds__(256) void core_bwd_dm_kernel(const float* __restrict__ dout, const float* __restrict__ Ar, float* __restrict__ dM,
                                                          int VQ, int A, int G, int K4, int rows_total) {
    // 8 rows (b, vq, g) per workgroup, 32 lanes per row: the row's A cotangents are loaded once and reused for all its float4 columns
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), c0 = threadIdx.x & 31;
    if (row >= rows_total) return;
    const int bvq = row / G, g = row - bvq * G, b = bvq / VQ;
    const float* dp = dout + (int64_t)bvq * A * G + g;
    float d[CB_AMAX];
#pragma unroll
    for (int a = 0; a < CB_AMAX; ++a) d[a] = a < A ? dp[a * G] : 0.f;
    const float4* ap = reinterpret_cast<const float4*>(Ar) + (int64_t)b * A * K4;
    float4* op = reinterpret_cast<float4*>(dM) + (int64_t)row * K4;
    for (int c = c0; c < K4; c += 32) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < CB_AMAX; ++a)
            if (a < A) {
                const float4 r = ap[(int64_t)a * K4 + c];
                o.x = fmaf(d[a], r.x, o.x); o.y = fmaf(d[a], r.y, o.y); o.z = fmaf(d[a], r.z, o.z); o.w = fmaf(d[a], r.w, o.w);
            }
        op[c] = o;
    }
}

// one workgroup per (sample, 128 columns): 32 column groups (float4) x 8 row phases; the phases meet in LDS in a fixed order
__global__ __launch_bounds__(256) void core_bwd_dar_kernel(const float* __restrict__ dout, const float* __restrict__ M, float* __restrict__ dAr,
                                                           int VQ, int A, int G, int K4) {
    __shared__ float4 red[8][CB_AMAX][32];
    const int b = blockIdx.y, cg = threadIdx.x & 31, rp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cg;
    const bool cok = c < K4;
    const int J = VQ * G;
    const float* db = dout + (int64_t)b * VQ * A * G;
    const float4* mb = reinterpret_cast<const float4*>(M) + (int64_t)b * J * K4 + c;
    float4 acc[CB_AMAX];
#pragma unroll
    for (int a = 0; a < CB_AMAX; ++a) acc[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = rp; j < J; j += 8) {
        const int vq = j / G, g = j - vq * G;
        const float4 m = cok ? mb[(int64_t)j * K4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* dp = db + (int64_t)vq * A * G + g;
#pragma unroll
        for (int a = 0; a < CB_AMAX; ++a)
            if (a < A) {
                const float d = dp[a * G];
                acc[a].x = fmaf(d, m.x, acc[a].x); acc[a].y = fmaf(d, m.y, acc[a].y); acc[a].z = fmaf(d, m.z, acc[a].z); acc[a].w = fmaf(d, m.w, acc[a].w);
            }
    }
#pragma unroll
    for (int a = 0; a < CB_AMAX; ++a)
        if (a < A) red[rp][a][cg] = acc[a];
    __syncthreads();
    for (int i = threadIdx.x; i < A * 32; i += 256) {
        const int a = i >> 5, cc = i & 31;
        if (blockIdx.x * 32 + cc < K4) {
            float4 s = red[0][a][cc];
#pragma unroll
            for (int r = 1; r < 8; ++r) { const float4 t = red[r][a][cc]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
            reinterpret_cast<float4*>(dAr)[((int64_t)b * A + a) * K4 + blockIdx.x * 32 + cc] = s;
        }
    }
}


// dAr with M held as bf16 hi/lo operand planes ([k >> 4][row][k & 15], the training forward's M): one workgroup per (16-column chunk, sample);
// a thread owns 8 columns of one row per step (one 16-B load per plane), 128 row phases; M = hi + lo (16 mantissa bits).  The sample's
// cotangents sit in LDS; the row phases meet by wave shuffles, the 4 waves in LDS, all in a fixed order.
__device__ __forceinline__ void unpack_bf16x8(const uint4 h, const uint4 l, float* m) {
    const uint32_t hh[4] = {h.x, h.y, h.z, h.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[2 * i] = __uint_as_float(hh[i] << 16) + __uint_as_float(ll[i] << 16);
        m[2 * i + 1] = __uint_as_float(hh[i] & 0xffff0000u) + __uint_as_float(ll[i] & 0xffff0000u);
    }
}
template <int AT>
__global__ __launch_bounds__(256) void core_bwd_dar_planes_kernel(const float* __restrict__ dout, const unsigned short* __restrict__ Mh,
                                                                  const unsigned short* __restrict__ Ml, int64_t rows_alloc, float* __restrict__ dAr,
                                                                  int VQ, int A, int G, int K, int lds_dout) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [4 waves][2][AT][8] partials, then dout[b] (VQ*A*G floats) when it fits
    float* red = sm;
    float* ds = sm + 4 * 2 * AT * 8;
    const int kc = blockIdx.x, b = blockIdx.y;
    const int t = threadIdx.x, kp = t & 1, rp = t >> 1, lane = t & 63, wid = t >> 6;
    const int J = VQ * G;
    const float* db = dout + (int64_t)b * VQ * A * G;
    if (lds_dout) {
        for (int i = t; i < VQ * A * G; i += 256) ds[i] = db[i];
        __syncthreads();
    }
    const float* dsrc = lds_dout ? ds : db;
    const int64_t base = ((int64_t)kc * rows_alloc + (int64_t)b * J) * 16 + kp * 8;
    const unsigned short* ph = Mh + base;
    const unsigned short* pl = Ml + base;
    float acc[AT][8];
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[a][e] = 0.f;
    for (int j0 = rp; j0 < J; j0 += 512) {                           // 4 rows per trip, their loads in flight together
        uint4 hv[4], lv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 128 * u;
            if (j < J) { hv[u] = *reinterpret_cast<const uint4*>(ph + (int64_t)j * 16); lv[u] = *reinterpret_cast<const uint4*>(pl + (int64_t)j * 16); }
            else { hv[u] = make_uint4(0, 0, 0, 0); lv[u] = hv[u]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 128 * u;
            if (j < J) {
                float m[8];
                unpack_bf16x8(hv[u], lv[u], m);
                const int vq = j / G, g = j - vq * G;
                const float* dp = dsrc + vq * A * G + g;
#pragma unroll
                for (int a = 0; a < AT; ++a)
                    if (a < A) {
                        const float d = dp[a * G];
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[a][e] = fmaf(d, m[e], acc[a][e]);
                    }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < AT; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[a][e];
#pragma unroll
            for (int o = 2; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);     // the 32 row phases of this wave (lane bit 0 is the column half)
            acc[a][e] = v;
        }
    if (lane < 2) {
#pragma unroll
        for (int a = 0; a < AT; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[((wid * 2 + lane) * AT + a) * 8 + e] = acc[a][e];
    }
    __syncthreads();
    for (int i = t; i < A * 16; i += 256) {
        const int a = i >> 4, kk = i & 15, k = kc * 16 + kk, half = kk >> 3, e = kk & 7;
        if (k < K) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[((w * 2 + half) * AT + a) * 8 + e];
            dAr[((int64_t)b * A + a) * K + k] = s;
        }
    }
}

}  // namespace
}  // namespace cti

using namespace cti;

extern "C" int cti_paralind_mbuild_bwd(const float* dM, const float* Vr, const float* Qr, const float* Teff, float* dVr, float* dQr,
                                       float* dTeff_partial, int B, int V, int Q, int R, int hr, int G, void* stream) {
    CTI_REQUIRE_PTR(dM); CTI_REQUIRE_PTR(Vr); CTI_REQUIRE_PTR(Qr); CTI_REQUIRE_PTR(Teff); CTI_REQUIRE_PTR(dVr); CTI_REQUIRE_PTR(dQr); CTI_REQUIRE_PTR(dTeff_partial);
    CTI_REQUIRE(B > 0 && B <= 65535 && V > 0 && Q > 0 && R > 0 && G > 0, CTI_E_SHAPE, "cti_paralind_mbuild_bwd: B=%d V=%d Q=%d R=%d G=%d", B, V, Q, R, G);
    CTI_REQUIRE(hr == 4 || hr == 8 || hr == 16, CTI_E_UNSUPPORTED, "cti_paralind_mbuild_bwd: h/rank=%d (built for 4, 8, 16)", hr);
    CTI_REQUIRE(Q * hr <= 1024 && hr * hr * G <= 1024 && V * G * hr <= 2048, CTI_E_UNSUPPORTED,
                "cti_paralind_mbuild_bwd: shape outside the kernel's per-thread budgets (Q*hr=%d, hr^2*G=%d, V*G*hr=%d)", Q * hr, hr * hr * G, V * G * hr);
    const size_t lds = sizeof(float) * ((size_t)hr * hr * hr * G + (size_t)V * G * hr * hr + (size_t)(V + 8) * hr + (size_t)Q * hr);
    int groups = (256 + B - 1) / B; if (groups > R) groups = R;
    const int rpb = (R + groups - 1) / groups;
    dim3 grid((R + rpb - 1) / rpb, B);
    int rc;
    {   // staged variant when its prefetch-register and LDS budgets hold (every model configuration)
        const int inner = hr * hr * G, rowsM = V * Q * G;
        const int region = hr * (inner + 8) > rowsM * (hr + 4) ? hr * (inner + 8) : rowsM * (hr + 4);
        const size_t lds_s = sizeof(float) * ((size_t)region + (size_t)V * G * hr * hr + (size_t)(V + 8) * hr + (size_t)Q * hr);
        if (hr * inner / 4 <= 2048 && rowsM * (hr / 4) <= 4096 && lds_s <= 160 * 1024) {
#define CTI_MBS(H) rc = set_lds(mbuild_bwd_staged_kernel<H>, lds_s, "cti_paralind_mbuild_bwd"); if (rc) return rc; \
    hipLaunchKernelGGL(mbuild_bwd_staged_kernel<H>, grid, dim3(1024), lds_s, as_stream(stream), dM, Vr, Qr, Teff, dVr, dQr, dTeff_partial, V, Q, R, G, rpb, region);
            if (hr == 4) { CTI_MBS(4) } else if (hr == 8) { CTI_MBS(8) } else { CTI_MBS(16) }
#undef CTI_MBS
            return launch_status("cti_paralind_mbuild_bwd");
        }
    }
#define CTI_MBB(H) rc = set_lds(mbuild_bwd_kernel<H>, lds, "cti_paralind_mbuild_bwd"); if (rc) return rc; \
    hipLaunchKernelGGL(mbuild_bwd_kernel<H>, grid, dim3(1024), lds, as_stream(stream), dM, Vr, Qr, Teff, dVr, dQr, dTeff_partial, V, Q, R, G, rpb);
    if (hr == 4) { CTI_MBB(4) } else if (hr == 8) { CTI_MBB(8) } else { CTI_MBB(16) }
#undef CTI_MBB
    return launch_status("cti_paralind_mbuild_bwd");
}

// Partials of dT_eff that cti_paralind_mbuild_bwd_mfma writes: one per chunk of samples (about 256 workgroups = R ranks x chunks), each chunk's sum formed in registers
extern "C" int cti_paralind_mbuild_bwd_mfma_partials(int B, int R) {
    if (B <= 0 || R <= 0) return 0;
    int nc = 256 / R; if (nc < 1) nc = 1; if (nc > B) nc = B;
    const int bpc = (B + nc - 1) / nc;
    return (B + bpc - 1) / bpc;
}

// The same gradients on the matrix cores (mbuild_bwd_mfma_kernel: bf16 hi / lo split products, fp32-grade; prec = CTI_PREC_BF16: one product per pair).
// CTI_E_UNSUPPORTED (nothing launched, no message) outside hr = 16, G = 2, V <= 48, Q <= 16, V*Q*G <= 1 024 or its LDS budget, and in the exact-fp32 mode:
// the caller takes cti_paralind_mbuild_bwd.
extern "C" int cti_paralind_mbuild_bwd_mfma(const float* dM, const float* Vr, const float* Qr, const float* Teff, float* dVr, float* dQr,
                                            float* dTeff_partial, int B, int V, int Q, int R, int hr, int G, int prec, void* stream) {
    CTI_REQUIRE_PTR(dM); CTI_REQUIRE_PTR(Vr); CTI_REQUIRE_PTR(Qr); CTI_REQUIRE_PTR(Teff); CTI_REQUIRE_PTR(dVr); CTI_REQUIRE_PTR(dQr); CTI_REQUIRE_PTR(dTeff_partial);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && R > 0 && R <= 65535 && G > 0, CTI_E_SHAPE, "cti_paralind_mbuild_bwd_mfma: B=%d V=%d Q=%d R=%d G=%d", B, V, Q, R, G);
    if (prec != CTI_PREC_BF16X3 && prec != CTI_PREC_BF16) return CTI_E_UNSUPPORTED;
    if (hr != 16 || G != 2 || V > 48 || Q > 16 || V * Q * G > 1024) return CTI_E_UNSUPPORTED;
    if (!aligned16b(dM) || !aligned16b(Teff)) return CTI_E_UNSUPPORTED;
    const int rowsM = V * Q * G;
    const int scratch = MBM_T_FLOATS + MBM_SCR_Q + MBM_SCR_V;
    const int region = rowsM * MBM_HP > scratch ? rowsM * MBM_HP : scratch;
    const size_t lds = sizeof(float) * ((size_t)region + (size_t)V * MBM_PV + 48 * 16 + 16 * 16);
    if (lds > 160 * 1024) return CTI_E_UNSUPPORTED;
    const int nparts = cti_paralind_mbuild_bwd_mfma_partials(B, R);
    const int bpc = (B + nparts - 1) / nparts;
    const dim3 grid(R, nparts);
    int rc;
    if (prec == CTI_PREC_BF16X3) {
        rc = set_lds(mbuild_bwd_mfma_kernel<3>, lds, "cti_paralind_mbuild_bwd_mfma"); if (rc) return rc;
        hipLaunchKernelGGL(mbuild_bwd_mfma_kernel<3>, grid, dim3(1024), lds, as_stream(stream), dM, Vr, Qr, Teff, dVr, dQr, dTeff_partial, B, V, Q, R, bpc, region);
    } else {
        rc = set_lds(mbuild_bwd_mfma_kernel<1>, lds, "cti_paralind_mbuild_bwd_mfma"); if (rc) return rc;
        hipLaunchKernelGGL(mbuild_bwd_mfma_kernel<1>, grid, dim3(1024), lds, as_stream(stream), dM, Vr, Qr, Teff, dVr, dQr, dTeff_partial, B, V, Q, R, bpc, region);
    }
    return launch_status("cti_paralind_mbuild_bwd_mfma");
}

// ---- M-build backward for ANY cubic core size (the reference takes any --rank / --h_mm, src/FFOE/main.py:61-64): plain fp32 VALU kernels,
// one thread per output element, through the same intermediates as the forward (X = mode-1 product):
//   X[b,v,r,j,k,g]  = sum_i T[r,i,j,k,g] Vr[b,v,r,i]          dX[b,v,r,j,k,g] = sum_q dM[b,v,q,g,r,k] Qr[b,q,r,j]
//   dQr[b,q,r,j]    = sum_{v,k,g} dM[b,v,q,g,r,k] X[b,v,r,j,k,g]
//   dVr[b,v,r,i]    = sum_{j,k,g} dX[b,v,r,j,k,g] T[r,i,j,k,g]   dT_partial[b][r,i,j,k,g] = sum_v Vr[b,v,r,i] dX[b,v,r,j,k,g]
// Exact fp32, slow (no tiling): the fallback behind the staged kernels built for h/rank in {4, 8, 16}.
namespace cti { namespace {
struct MbgP { const float* dM; const float* Vr; const float* Qr; const float* T; float* X; float* dX; float* dVr; float* dQr; float* dTp; int B, V, Q, R, H, G; };
__global__ __launch_bounds__(256) void mbg_x_dx_kernel(MbgP p) {               // thread = (b, v, r, j, k, g)
    const int64_t n = (int64_t)p.B * p.V * p.R * p.H * p.H * p.G, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int64_t u = t;
    const int g = u % p.G; u /= p.G; const int k = u % p.H; u /= p.H; const int j = u % p.H; u /= p.H; const int r = u % p.R; u /= p.R;
    const int v = u % p.V; const int b = u / p.V;
    const int K = p.R * p.H;
    float x = 0.f, dx = 0.f;
    const float* vr = p.Vr + ((int64_t)b * p.V + v) * K + r * p.H;
    for (int i = 0; i < p.H; ++i) x = fmaf(p.T[((((int64_t)r * p.H + i) * p.H + j) * p.H + k) * p.G + g], vr[i], x);
    for (int q = 0; q < p.Q; ++q)
        dx = fmaf(p.dM[((((int64_t)b * p.V + v) * p.Q + q) * p.G + g) * K + r * p.H + k], p.Qr[((int64_t)b * p.Q + q) * K + r * p.H + j], dx);
    p.X[t] = x; p.dX[t] = dx;
}
__global__ __launch_bounds__(256) void mbg_dq_kernel(MbgP p) {                 // thread = (b, q, r, j)
    const int64_t n = (int64_t)p.B * p.Q * p.R * p.H, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int64_t u = t;
    const int j = u % p.H; u /= p.H; const int r = u % p.R; u /= p.R; const int q = u % p.Q; const int b = u / p.Q;
    const int K = p.R * p.H;
    float s = 0.f;
    for (int v = 0; v < p.V; ++v)
        for (int k = 0; k < p.H; ++k)
            for (int g = 0; g < p.G; ++g)
                s = fmaf(p.dM[((((int64_t)b * p.V + v) * p.Q + q) * p.G + g) * K + r * p.H + k],
                         p.X[(((((int64_t)b * p.V + v) * p.R + r) * p.H + j) * p.H + k) * p.G + g], s);
    p.dQr[((int64_t)b * p.Q + q) * K + r * p.H + j] = s;
}
__global__ __launch_bounds__(256) void mbg_dv_kernel(MbgP p) {                 // thread = (b, v, r, i)
    const int64_t n = (int64_t)p.B * p.V * p.R * p.H, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int64_t u = t;
    const int i = u % p.H; u /= p.H; const int r = u % p.R; u /= p.R; const int v = u % p.V; const int b = u / p.V;
    const int inner = p.H * p.H * p.G;
    const float* dx = p.dX + (((int64_t)b * p.V + v) * p.R + r) * inner;
    const float* tt = p.T + ((int64_t)r * p.H + i) * inner;
    float s = 0.f;
    for (int c = 0; c < inner; ++c) s = fmaf(dx[c], tt[c], s);
    p.dVr[((int64_t)b * p.V + v) * (p.R * p.H) + r * p.H + i] = s;
}
__global__ __launch_bounds__(256) void mbg_dt_kernel(MbgP p) {                 // thread = (b, r, i, c = (j,k,g))
    const int inner = p.H * p.H * p.G;
    const int64_t n = (int64_t)p.B * p.R * p.H * inner, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int64_t u = t;
    const int c = u % inner; u /= inner; const int i = u % p.H; u /= p.H; const int r = u % p.R; const int b = u / p.R;
    float s = 0.f;
    for (int v = 0; v < p.V; ++v)
        s = fmaf(p.Vr[((int64_t)b * p.V + v) * (p.R * p.H) + r * p.H + i], p.dX[(((int64_t)b * p.V + v) * p.R + r) * inner + c], s);
    p.dTp[t] = s;
}
} }
extern "C" size_t cti_paralind_mbuild_bwd_generic_workspace_bytes(int B, int V, int R, int hr, int G) {
    if (B <= 0 || V <= 0 || R <= 0 || hr <= 0 || G <= 0) return 0;
    return 2 * sizeof(float) * (size_t)B * V * R * hr * hr * G;
}
extern "C" int cti_paralind_mbuild_bwd_generic(const float* dM, const float* Vr, const float* Qr, const float* Teff, float* dVr, float* dQr,
                                               float* dTeff_partial, int B, int V, int Q, int R, int hr, int G, void* workspace, size_t workspace_bytes,
                                               void* stream) {
    CTI_REQUIRE_PTR(dM); CTI_REQUIRE_PTR(Vr); CTI_REQUIRE_PTR(Qr); CTI_REQUIRE_PTR(Teff); CTI_REQUIRE_PTR(dVr); CTI_REQUIRE_PTR(dQr); CTI_REQUIRE_PTR(dTeff_partial);
    CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && R > 0 && hr > 0 && G > 0, CTI_E_SHAPE, "cti_paralind_mbuild_bwd_generic: B=%d V=%d Q=%d R=%d hr=%d G=%d", B, V, Q, R, hr, G);
    CTI_REQUIRE(workspace_bytes >= cti_paralind_mbuild_bwd_generic_workspace_bytes(B, V, R, hr, G), CTI_E_WORKSPACE, "cti_paralind_mbuild_bwd_generic: workspace too small");
    const int64_t nx = (int64_t)B * V * R * hr * hr * G;
    CTI_REQUIRE((nx + 255) / 256 < 0x7fffffffLL && ((int64_t)B * R * hr * hr * hr * G + 255) / 256 < 0x7fffffffLL, CTI_E_SHAPE, "cti_paralind_mbuild_bwd_generic: too large");
    MbgP p{dM, Vr, Qr, Teff, static_cast<float*>(workspace), static_cast<float*>(workspace) + nx, dVr, dQr, dTeff_partial, B, V, Q, R, hr, G};
    hipStream_t st = as_stream(stream);
    auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    hipLaunchKernelGGL(mbg_x_dx_kernel, blocks(nx), dim3(256), 0, st, p);
    hipLaunchKernelGGL(mbg_dq_kernel, blocks((int64_t)B * Q * R * hr), dim3(256), 0, st, p);
    hipLaunchKernelGGL(mbg_dv_kernel, blocks((int64_t)B * V * R * hr), dim3(256), 0, st, p);
    hipLaunchKernelGGL(mbg_dt_kernel, blocks((int64_t)B * R * hr * hr * hr * G), dim3(256), 0, st, p);
    return launch_status("cti_paralind_mbuild_bwd_generic");
}

static void tri_chunks_b(int V, int64_t QA, int64_t* chunk_n, int* nchunk) {
    const int64_t N = (int64_t)V * QA, c = tuning_tri_chunk() > 0 ? tuning_tri_chunk() : 32768;
    *nchunk = (int)((N + c - 1) / c); *chunk_n = c;
}
extern "C" size_t cti_softmax_tri_bwd_workspace_bytes(int B, int V, int64_t QA, int G) {
    if (B <= 0 || V <= 0 || QA <= 0 || G <= 0) return 0;
    int64_t cn; int nc; tri_chunks_b(V, QA, &cn, &nc);
    return sizeof(float) * (size_t)B * nc * G;
}
extern "C" int cti_masked_softmax_tri_bwd(const float* p, const float* dp, float* dlogits, int B, int V, int64_t QA, int G, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(p); CTI_REQUIRE_PTR(dp); CTI_REQUIRE_PTR(dlogits); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(B > 0 && B <= 65535 && V > 0 && QA > 0 && G > 0 && G <= 64, CTI_E_SHAPE, "cti_masked_softmax_tri_bwd: B=%d V=%d QA=%lld G=%d", B, V, (long long)QA, G);
    CTI_REQUIRE(workspace_bytes >= cti_softmax_tri_bwd_workspace_bytes(B, V, QA, G), CTI_E_WORKSPACE, "cti_masked_softmax_tri_bwd: workspace too small");
    int64_t cn; int nc; tri_chunks_b(V, QA, &cn, &nc);
    const int64_t N = (int64_t)V * QA, NG = N * G;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(tri_sm_bwd_partial, dim3(nc, B), dim3(256), 0, as_stream(stream), p, dp, part, N, G, cn, nc);
    int rc = launch_status("cti_masked_softmax_tri_bwd/partial"); if (rc) return rc;
    const int64_t nblk = (NG + 255) / 256;
    hipLaunchKernelGGL(tri_sm_bwd_apply, dim3((unsigned)(nblk < 2048 ? nblk : 2048), B), dim3(256), 0, as_stream(stream), p, dp, part, dlogits, NG, G, nc);
    return launch_status("cti_masked_softmax_tri_bwd/apply");
}
extern "C" int cti_masked_softmax_bi_bwd(const float* p, const float* dp, float* dlogits, int rows, int N, void* stream) {
    CTI_REQUIRE_PTR(p); CTI_REQUIRE_PTR(dp); CTI_REQUIRE_PTR(dlogits);
    CTI_REQUIRE(rows > 0 && N > 0, CTI_E_SHAPE, "cti_masked_softmax_bi_bwd: rows=%d N=%d", rows, N);
    hipLaunchKernelGGL(bi_sm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), p, dp, dlogits, rows, N);
    return launch_status("cti_masked_softmax_bi_bwd");
}

extern "C" int cti_tri_pool_bwd(const float* dout, const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                                int64_t w_sq, int64_t w_sa, float* dvt, float* dqt, float* dat, float* dw, int B, int V, int Q, int A, int D,
                                void* stream) {
    CTI_REQUIRE_PTR(dout); CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(at); CTI_REQUIRE_PTR(w);
    CTI_REQUIRE_PTR(dvt); CTI_REQUIRE_PTR(dqt); CTI_REQUIRE_PTR(dat);
    CTI_REQUIRE(B > 0 && B <= 65535 && V > 0 && Q > 0 && A > 0 && D > 0, CTI_E_SHAPE, "cti_tri_pool_bwd: B=%d V=%d Q=%d A=%d D=%d", B, V, Q, A, D);
    int rc;
    const int AP = A <= 4 ? 4 : 8;
    if (A <= 8 && Q <= 16 && sizeof(float) * (size_t)V * Q * AP <= 64 * 1024) {
        const size_t lds_s = sizeof(float) * (size_t)V * Q * AP;
        if (AP == 4) hipLaunchKernelGGL(tri_pool_bwd_small_kernel<4>, dim3((D + 255) / 256, B), dim3(256), lds_s, as_stream(stream), dout, vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, dvt, dqt, dat, V, Q, A, D);
        else         hipLaunchKernelGGL(tri_pool_bwd_small_kernel<8>, dim3((D + 255) / 256, B), dim3(256), lds_s, as_stream(stream), dout, vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, dvt, dqt, dat, V, Q, A, D);
    } else {
        const size_t lds = sizeof(float) * 256 * 2 * (size_t)(Q + A);
        rc = set_lds(tri_pool_bwd_kernel, lds, "cti_tri_pool_bwd"); if (rc) return rc;
        hipLaunchKernelGGL(tri_pool_bwd_kernel, dim3((D + 255) / 256, B), dim3(256), lds, as_stream(stream), dout, vt, qt, at, w, w_sb, w_sv, w_sq, w_sa,
                           dvt, dqt, dat, V, Q, A, D);
    }
    rc = launch_status("cti_tri_pool_bwd"); if (rc) return rc;
    if (dw) {
        // (q, a) accumulator block per lane: A <= 4 (the FFOE model's 3 answer tokens) -> 8 x 4, no FMAs on padded answers and half the sweeps over D
        if (A <= 4) hipLaunchKernelGGL((pool_dw_kernel<8, 4>), dim3((unsigned)(B * V)), dim3(256), 0, as_stream(stream), dout, vt, qt, at, dw, V, Q, A, D, 1);
        else        hipLaunchKernelGGL((pool_dw_kernel<4, 8>), dim3((unsigned)(B * V)), dim3(256), 0, as_stream(stream), dout, vt, qt, at, dw, V, Q, A, D, 1);
        rc = launch_status("cti_tri_pool_bwd/dw");
    }
    return rc;
}

extern "C" int cti_bi_pool_bwd(const float* dout, const float* vt, const float* qt, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                               float* dvt, float* dqt, float* dw, int B, int V, int Q, int D, int k, void* stream) {
    CTI_REQUIRE_PTR(dout); CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(dvt); CTI_REQUIRE_PTR(dqt);
    CTI_REQUIRE(B > 0 && B <= 65535 && V > 0 && Q > 0 && D > 0 && k > 0 && D >= k, CTI_E_SHAPE, "cti_bi_pool_bwd: B=%d V=%d Q=%d D=%d k=%d", B, V, Q, D, k);
    int rc;
    const uintptr_t al8 = reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(vt) | reinterpret_cast<uintptr_t>(qt) | reinterpret_cast<uintptr_t>(dvt) |
                          reinterpret_cast<uintptr_t>(dqt);
    if (k == 1 && w && Q <= 16 && D % 2 == 0 && (al8 & 7) == 0 && (size_t)V * 16 * sizeof(float) <= 48 * 1024) {
        hipLaunchKernelGGL(bi_pool_bwd_reg_kernel, dim3((D / 2 + 127) / 128, B), dim3(128), sizeof(float) * (size_t)V * 16, as_stream(stream), dout, vt, qt, w,
                           w_sb, w_sv, w_sq, dvt, dqt, V, Q, D);
    } else {
        const size_t lds = sizeof(float) * 256 * 2 * (size_t)Q;
        rc = set_lds(bi_pool_bwd_kernel, lds, "cti_bi_pool_bwd"); if (rc) return rc;
        hipLaunchKernelGGL(bi_pool_bwd_kernel, dim3((D + 255) / 256, B), dim3(256), lds, as_stream(stream), dout, vt, qt, w, w_sb, w_sv, w_sq, dvt, dqt,
                           V, Q, D, k);
    }
    rc = launch_status("cti_bi_pool_bwd"); if (rc) return rc;
    if (dw && w) {
        const int Du = (D / k) * k;                         // channels of a ragged tail take no part
        hipLaunchKernelGGL((pool_dw_kernel<16, 1>), dim3((unsigned)(B * V)), dim3(256), 0, as_stream(stream), dout, vt, qt, (const float*)nullptr, dw, V, Q, 1, Du, k);
        rc = launch_status("cti_bi_pool_bwd/dw");
    }
    return rc;
}

extern "C" int cti_bi_logits_bwd(const float* dlogits, const float* vt, const float* qt, const float* h, const float* h_scale, float* dvt,
                                 float* dqt, float* dh_partial, float* dh_bias_partial, int B, int G, int V, int Q, int D, void* stream) {
    CTI_REQUIRE_PTR(dlogits); CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(h); CTI_REQUIRE_PTR(dvt); CTI_REQUIRE_PTR(dqt);
    CTI_REQUIRE_PTR(dh_partial); CTI_REQUIRE_PTR(dh_bias_partial);
    CTI_REQUIRE(B > 0 && B <= 65535 && G > 0 && V > 0 && Q > 0 && D > 0, CTI_E_SHAPE, "cti_bi_logits_bwd: B=%d G=%d V=%d Q=%d D=%d", B, G, V, Q, D);
    const size_t lds = sizeof(float) * 256 * 2 * (size_t)(V + Q);
    int rc = set_lds(bi_logits_bwd_kernel, lds, "cti_bi_logits_bwd"); if (rc) return rc;
    hipLaunchKernelGGL(bi_logits_bwd_kernel, dim3((D + 255) / 256, B), dim3(256), lds, as_stream(stream), dlogits, vt, qt, h, h_scale, dvt, dqt,
                       dh_partial, G, V, Q, D);
    rc = launch_status("cti_bi_logits_bwd"); if (rc) return rc;
    const int64_t rows = (int64_t)B * G;                    // dh_bias_partial[b,g] = sum_{v,q} dlogits[b,g,v,q]
    hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), dlogits, dh_bias_partial, rows, V * Q);
    return launch_status("cti_bi_logits_bwd/bias");
}

extern "C" int cti_row_sum(const float* x, float* out, int64_t rows, int cols, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(rows > 0 && cols > 0, CTI_E_SHAPE, "cti_row_sum: rows=%lld cols=%d", (long long)rows, cols);
    hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), x, out, rows, cols);
    return launch_status("cti_row_sum");
}

extern "C" int cti_paralind_core_bwd(const float* dout, const float* M, const float* Ar, float* dM, float* dAr, int B, int V, int Q, int A, int G,
                                     int K, void* stream) {
    CTI_REQUIRE_PTR(dout); CTI_REQUIRE_PTR(M); CTI_REQUIRE_PTR(Ar); CTI_REQUIRE_PTR(dM); CTI_REQUIRE_PTR(dAr);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && A > 0 && G > 0 && K > 0, CTI_E_SHAPE, "cti_paralind_core_bwd: B=%d V=%d Q=%d A=%d G=%d K=%d", B, V, Q, A, G, K);
    const uintptr_t al = reinterpret_cast<uintptr_t>(M) | reinterpret_cast<uintptr_t>(Ar) | reinterpret_cast<uintptr_t>(dM) | reinterpret_cast<uintptr_t>(dAr);
    if (K % 4 != 0 || A > CB_AMAX || B > 65535 || (al & 15)) return CTI_E_UNSUPPORTED;      // the caller takes the transposed-GEMM route
    const int K4 = K / 4, VQ = V * Q;
    const int64_t total4 = (int64_t)B * VQ * G * K4;
    CTI_REQUIRE((total4 + 255) / 256 <= 0x7fffffffLL, CTI_E_SHAPE, "cti_paralind_core_bwd: %lld elements exceed one launch", (long long)total4 * 4);
    CTI_REQUIRE(total4 / K4 <= 0x7fffffffLL, CTI_E_SHAPE, "cti_paralind_core_bwd: %lld rows exceed one launch", (long long)(total4 / K4));
    hipLaunchKernelGGL(core_bwd_dm_kernel, dim3((unsigned)((total4 / K4 + 7) / 8)), dim3(256), 0, as_stream(stream), dout, Ar, dM, VQ, A, G, K4, (int)(total4 / K4));
    int rc = launch_status("cti_paralind_core_bwd/dM"); if (rc) return rc;
    hipLaunchKernelGGL(core_bwd_dar_kernel, dim3((K4 + 31) / 32, B), dim3(256), 0, as_stream(stream), dout, M, dAr, VQ, A, G, K4);
    return launch_status("cti_paralind_core_bwd/dAr");
}

extern "C" int cti_paralind_core_bwd_planes(const float* dout, const void* Mh, const void* Ml, int64_t rows_alloc, const float* Ar, float* dM, float* dAr,
                                            int B, int V, int Q, int A, int G, int K, void* stream) {
    CTI_REQUIRE_PTR(dout); CTI_REQUIRE_PTR(Mh); CTI_REQUIRE_PTR(Ml); CTI_REQUIRE_PTR(Ar); CTI_REQUIRE_PTR(dM); CTI_REQUIRE_PTR(dAr);
    CTI_REQUIRE(B > 0 && B <= 65535 && V > 0 && Q > 0 && A > 0 && A <= CB_AMAX && G > 0 && K > 0 && K % 32 == 0 &&
                rows_alloc >= (int64_t)B * V * Q * G, CTI_E_SHAPE, "cti_paralind_core_bwd_planes: B=%d V=%d Q=%d A=%d (<= 8) G=%d K=%d (a multiple of 32)", B, V, Q, A, G, K);
    CTI_REQUIRE(((reinterpret_cast<uintptr_t>(Ar) | reinterpret_cast<uintptr_t>(dM) | reinterpret_cast<uintptr_t>(Mh) | reinterpret_cast<uintptr_t>(Ml)) & 15) == 0,
                CTI_E_SHAPE, "cti_paralind_core_bwd_planes: operands must be 16-B aligned");
    const int K4 = K / 4, VQ = V * Q;
    const int64_t total4 = (int64_t)B * VQ * G * K4;
    CTI_REQUIRE((total4 + 255) / 256 <= 0x7fffffffLL, CTI_E_SHAPE, "cti_paralind_core_bwd_planes: %lld elements exceed one launch", (long long)total4 * 4);
    CTI_REQUIRE(total4 / K4 <= 0x7fffffffLL, CTI_E_SHAPE, "cti_paralind_core_bwd_planes: %lld rows exceed one launch", (long long)(total4 / K4));
    hipLaunchKernelGGL(core_bwd_dm_kernel, dim3((unsigned)((total4 / K4 + 7) / 8)), dim3(256), 0, as_stream(stream), dout, Ar, dM, VQ, A, G, K4, (int)(total4 / K4));
    int rc = launch_status("cti_paralind_core_bwd_planes/dM"); if (rc) return rc;
    const size_t dbytes = sizeof(float) * (size_t)VQ * A * G;
    const int lds_dout = dbytes <= 48 * 1024;
    const unsigned short* mh = static_cast<const unsigned short*>(Mh);
    const unsigned short* ml = static_cast<const unsigned short*>(Ml);
#define CTI_DAR(ATv) hipLaunchKernelGGL((core_bwd_dar_planes_kernel<ATv>), dim3(K / 16, B), dim3(256), sizeof(float) * 4 * 2 * ATv * 8 + (lds_dout ? dbytes : 0), \
                                        as_stream(stream), dout, mh, ml, rows_alloc, dAr, VQ, A, G, K, lds_dout)
    if (A <= 3) CTI_DAR(3); else if (A <= 6) CTI_DAR(6); else CTI_DAR(8);
#undef CTI_DAR
    return launch_status("cti_paralind_core_bwd_planes/dAr");
}
