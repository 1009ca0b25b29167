// cti_attention.hip -- masked softmax (Tri / Bi), attention-weighted sum-pools (tri / bi) and the bilinear
// attention logits.  HBM-bound byte movers: the lane axis is always the contiguous axis of the largest operand,
// reductions are wave shuffles (64 lanes) + a small LDS tree, nothing is reshaped into a GEMM.
#include "cti_common.h"
#include <math.h>

namespace cti {
namespace {

__device__ __forceinline__ float neg_inf() { return -__builtin_huge_valf(); }

// =====================================================================================================
// Tri softmax.  logits (B, N = V*QA, G), G innermost; softmax over N per (b, g)      (attention.py:55-58)
// pass 1: per (b, chunk): fill -inf on masked rows, chunk max m_c[g] and s_c[g] = sum exp(x - m_c)
// pass 2: per b: m = max_c m_c, S = sum_c s_c * exp(m_c - m)  ->  stats[b][g] = (m, S)
// pass 3: p = exp(x - m) / S
// Every thread walks flat indices f = e*G + g of its chunk with stride 256; lane l's g is fixed iff
// 256 % G == 0, which is not assumed: per-g accumulators live in LDS-free registers for G <= GMAX via select.
// =====================================================================================================
constexpr int GMAX = 8;          // glimpse <= 8 in every reference configuration; larger G uses the slow path below
constexpr int SM_THREADS = 256;

template <int GT>  // GT = compile-time G (1..8) or 0 = generic (one g per pass)
__global__ __launch_bounds__(SM_THREADS) void tri_partial_kernel(float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                                 float* __restrict__ part /* [B][nchunk][G][2] */,
                                                                 int V, int64_t QA, int G, int64_t chunk_n, int nchunk) {
    __shared__ float red[2][SM_THREADS / 64][GMAX];
    const int b = blockIdx.y, c = blockIdx.x, t = threadIdx.x;
    const int64_t N = (int64_t)V * QA;
    const int64_t n_lo = (int64_t)c * chunk_n, n_hi = min(N, n_lo + chunk_n);
    float* x = logits + (int64_t)b * N * G;
    const uint8_t* mk = mask + (int64_t)b * V;
    const int Gr = GT ? GT : G;

    for (int g0 = 0; g0 < Gr; g0 += GMAX) {            // one trip unless G > 8
        float mx[GMAX], sm[GMAX];
#pragma unroll
        for (int g = 0; g < GMAX; ++g) { mx[g] = neg_inf(); sm[g] = 0.f; }
        // pass A: mask fill + max
        for (int64_t n = n_lo + t; n < n_hi; n += SM_THREADS) {
            const bool masked = mk[n / QA] != 0;
            float* row = x + n * Gr;
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                if (g0 + g < Gr) {
                    float v = row[g0 + g];
                    if (masked) { v = neg_inf(); row[g0 + g] = v; }
                    mx[g] = fmaxf(mx[g], v);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < GMAX; ++g) mx[g] = wave_max(mx[g]);
        if ((t & 63) == 0) {
#pragma unroll
            for (int g = 0; g < GMAX; ++g) red[0][t >> 6][g] = mx[g];
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            float m = red[0][0][g];
#pragma unroll
            for (int w = 1; w < SM_THREADS / 64; ++w) m = fmaxf(m, red[0][w][g]);
            mx[g] = m;
        }
        // pass B: sum exp(x - m_c) (chunk re-read comes from L2: a chunk is <= 256 KiB)
        for (int64_t n = n_lo + t; n < n_hi; n += SM_THREADS) {
            const float* row = x + n * Gr;
#pragma unroll
            for (int g = 0; g < GMAX; ++g) {
                if (g0 + g < Gr) {
                    const float v = row[g0 + g];
                    if (mx[g] != neg_inf()) sm[g] += expf(v - mx[g]);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < GMAX; ++g) sm[g] = wave_sum(sm[g]);
        if ((t & 63) == 0) {
#pragma unroll
            for (int g = 0; g < GMAX; ++g) red[1][t >> 6][g] = sm[g];
        }
        __syncthreads();
        if (t < GMAX && g0 + t < Gr) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < SM_THREADS / 64; ++w) s += red[1][w][t];
            float* o = part + (((int64_t)b * nchunk + c) * Gr + g0 + t) * 2;
            o[0] = red[0][0][t];
#pragma unroll
            for (int w = 1; w < SM_THREADS / 64; ++w) o[0] = fmaxf(o[0], red[0][w][t]);
            o[1] = s;
        }
        __syncthreads();
    }
}

// G = 2 fast path of the partial pass (every TriAttention of the reference's models): a thread handles TWO positions per 16-B load,
// the owning object v = n / QA is tracked incrementally (the generic kernel pays a 64-bit division per element), and max / sum are
// accumulated in ONE sweep (thread-local running maximum, rescaled when it moves), so the chunk is read once.
__global__ __launch_bounds__(SM_THREADS) void tri_partial_g2_kernel(float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                                    float* __restrict__ part, int V, int64_t QA, int64_t chunk_n, int nchunk) {
    __shared__ float red[2][SM_THREADS / 64][2];
    const int b = blockIdx.y, c = blockIdx.x, t = threadIdx.x;
    const int64_t N = (int64_t)V * QA;
    const int64_t n_lo = (int64_t)c * chunk_n, n_hi = min(N, n_lo + chunk_n);
    float* x = logits + (int64_t)b * N * 2;
    const uint8_t* mk = mask + (int64_t)b * V;
    float m0 = neg_inf(), m1 = neg_inf(), s0 = 0.f, s1 = 0.f;
    auto upd = [](float v, float& m, float& sacc) {
        if (v > m) { sacc = sacc * __expf(m - v) + 1.f; m = v; }           // m = -inf: exp(-inf) = 0
        else if (v != neg_inf()) sacc += __expf(v - m);
    };
    // positions n, n+1 per thread and trip (n even: n_lo and chunk_n are even, N even or the tail is handled below)
    int64_t n = n_lo + 2 * t;
    int64_t vcur = n < n_hi ? n / QA : 0;
    int64_t bound = (vcur + 1) * QA;                                       // first position of the next object
    for (; n + 1 < n_hi; n += 2 * SM_THREADS) {
        while (n >= bound) { ++vcur; bound += QA; }
        const bool ma = mk[vcur] != 0;
        const bool mb = (n + 1 >= bound) ? (mk[vcur + 1] != 0) : ma;
        float4 v4 = *reinterpret_cast<const float4*>(x + n * 2);
        if (ma) { v4.x = neg_inf(); v4.y = neg_inf(); }
        if (mb) { v4.z = neg_inf(); v4.w = neg_inf(); }
        if (ma || mb) *reinterpret_cast<float4*>(x + n * 2) = v4;
        upd(v4.x, m0, s0); upd(v4.y, m1, s1); upd(v4.z, m0, s0); upd(v4.w, m1, s1);
    }
    if (n < n_hi) {                                                        // odd tail: one position
        while (n >= bound) { ++vcur; bound += QA; }
        float2 v2 = *reinterpret_cast<const float2*>(x + n * 2);
        if (mk[vcur] != 0) { v2.x = neg_inf(); v2.y = neg_inf(); *reinterpret_cast<float2*>(x + n * 2) = v2; }
        upd(v2.x, m0, s0); upd(v2.y, m1, s1);
    }
    // combine the threads: common maximum, rescaled sums
    const float w0 = wave_max(m0), w1 = wave_max(m1);
    s0 = wave_sum(m0 == neg_inf() ? 0.f : s0 * __expf(m0 - w0));
    s1 = wave_sum(m1 == neg_inf() ? 0.f : s1 * __expf(m1 - w1));
    if ((t & 63) == 0) { red[0][t >> 6][0] = w0; red[0][t >> 6][1] = w1; red[1][t >> 6][0] = s0; red[1][t >> 6][1] = s1; }
    __syncthreads();
    if (t < 2) {
        float m = red[0][0][t];
#pragma unroll
        for (int w = 1; w < SM_THREADS / 64; ++w) m = fmaxf(m, red[0][w][t]);
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < SM_THREADS / 64; ++w) if (red[0][w][t] != neg_inf()) sacc += red[1][w][t] * __expf(red[0][w][t] - m);
        float* o = part + (((int64_t)b * nchunk + c) * 2 + t) * 2;
        o[0] = m; o[1] = sacc;
    }
}

// p = exp(x - m_g) / s_g, four elements per thread when G divides 4 (g of element u is u % G: the float4 starts at a multiple of 4)
__global__ __launch_bounds__(256) void tri_normalise_v4_kernel(const float* __restrict__ logits, const float* __restrict__ stats,
                                                               float* __restrict__ p, int64_t NG4, int G) {
    const int b = blockIdx.y;
    const float4* x = reinterpret_cast<const float4*>(logits) + (int64_t)b * NG4;
    float4* y = reinterpret_cast<float4*>(p) + (int64_t)b * NG4;
    const float* st = stats + (int64_t)b * G * 2;
    float m[4], inv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { m[u] = st[(u % G) * 2]; inv[u] = 1.f / st[(u % G) * 2 + 1]; }
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < NG4; f += stride) {
        const float4 v = x[f];
        y[f] = make_float4(__expf(v.x - m[0]) * inv[0], __expf(v.y - m[1]) * inv[1], __expf(v.z - m[2]) * inv[2], __expf(v.w - m[3]) * inv[3]);
    }
}

// G = 2 normalise pass that also applies the mask (the partial pass ran in the GEMM's epilogue and left the logits untouched): masked
// rows get -inf in `logits` (the reference's in-place masked_fill_) and exp(-inf - m) = 0 in p.  Two positions per thread and 16-B access,
// the owning object v = n / QA tracked incrementally as in tri_partial_g2_kernel.
__global__ __launch_bounds__(SM_THREADS) void tri_normalise_mask_g2_kernel(float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                                          const float* __restrict__ stats, float* __restrict__ p,
                                                                          int V, int64_t QA, int64_t chunk_n) {
    const int b = blockIdx.y, c = blockIdx.x, t = threadIdx.x;
    const int64_t N = (int64_t)V * QA;
    const int64_t n_lo = (int64_t)c * chunk_n, n_hi = min(N, n_lo + chunk_n);
    float* x = logits + (int64_t)b * N * 2;
    float* y = p + (int64_t)b * N * 2;
    const uint8_t* mk = mask + (int64_t)b * V;
    const float* st = stats + (int64_t)b * 4;
    constexpr float L2E = 1.4426950408889634f;
    // exp(x - m) / S = exp2(x * log2(e) - m * log2(e)) * (1 / S): one FMA and one v_exp_f32 per element
    const float m0 = st[0] * L2E, i0 = 1.f / st[1], m1 = st[2] * L2E, i1 = 1.f / st[3];
    auto pr = [&](float4 v4) {
        return make_float4(__builtin_amdgcn_exp2f(fmaf(v4.x, L2E, -m0)) * i0, __builtin_amdgcn_exp2f(fmaf(v4.y, L2E, -m1)) * i1,
                           __builtin_amdgcn_exp2f(fmaf(v4.z, L2E, -m0)) * i0, __builtin_amdgcn_exp2f(fmaf(v4.w, L2E, -m1)) * i1);
    };
    // two 16-B loads in flight per thread: positions (n, n+1) and (n + 512, n + 513)
    int64_t n = n_lo + 2 * t;
    int64_t va = n < n_hi ? n / QA : 0, ba = (va + 1) * QA;                        // object of the first pair and the first position of the next object
    int64_t vb = n + 2 * SM_THREADS < n_hi ? (n + 2 * SM_THREADS) / QA : 0, bb = (vb + 1) * QA;
    for (; n + 2 * SM_THREADS + 1 < n_hi; n += 4 * SM_THREADS) {
        const int64_t n2 = n + 2 * SM_THREADS;
        while (n >= ba) { ++va; ba += QA; }
        while (n2 >= bb) { ++vb; bb += QA; }
#ifndef CTI_SM_NT
#define CTI_SM_NT 0        // 1: non-temporal loads and stores (measured: no difference, 351 vs 347 us at B = 64)
#endif
#if CTI_SM_NT
        typedef float nf4 __attribute__((ext_vector_type(4)));
        const nf4 un = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(x + n * 2)), wn = __builtin_nontemporal_load(reinterpret_cast<const nf4*>(x + n2 * 2));
        float4 u = make_float4(un[0], un[1], un[2], un[3]), w = make_float4(wn[0], wn[1], wn[2], wn[3]);
#else
        float4 u = *reinterpret_cast<const float4*>(x + n * 2), w = *reinterpret_cast<const float4*>(x + n2 * 2);
#endif
        const bool ua = mk[va] != 0, ub = (n + 1 >= ba) ? (mk[va + 1] != 0) : ua;
        const bool wa = mk[vb] != 0, wb = (n2 + 1 >= bb) ? (mk[vb + 1] != 0) : wa;
        if (ua) { u.x = neg_inf(); u.y = neg_inf(); }
        if (ub) { u.z = neg_inf(); u.w = neg_inf(); }
        if (wa) { w.x = neg_inf(); w.y = neg_inf(); }
        if (wb) { w.z = neg_inf(); w.w = neg_inf(); }
        if (ua || ub) *reinterpret_cast<float4*>(x + n * 2) = u;
        if (wa || wb) *reinterpret_cast<float4*>(x + n2 * 2) = w;
#if CTI_SM_NT
        { const float4 a = pr(u), b2 = pr(w); nf4 an = {a.x, a.y, a.z, a.w}, bn = {b2.x, b2.y, b2.z, b2.w};
          __builtin_nontemporal_store(an, reinterpret_cast<nf4*>(y + n * 2)); __builtin_nontemporal_store(bn, reinterpret_cast<nf4*>(y + n2 * 2)); }
#else
        *reinterpret_cast<float4*>(y + n * 2) = pr(u);
        *reinterpret_cast<float4*>(y + n2 * 2) = pr(w);
#endif
    }
    for (; n < n_hi; n += 2 * SM_THREADS) {                                        // the chunk's tail: one pair (or one position) per trip
        while (n >= ba) { ++va; ba += QA; }
        if (n + 1 < n_hi) {
            float4 u = *reinterpret_cast<const float4*>(x + n * 2);
            const bool ua = mk[va] != 0, ub = (n + 1 >= ba) ? (mk[va + 1] != 0) : ua;
            if (ua) { u.x = neg_inf(); u.y = neg_inf(); }
            if (ub) { u.z = neg_inf(); u.w = neg_inf(); }
            if (ua || ub) *reinterpret_cast<float4*>(x + n * 2) = u;
            *reinterpret_cast<float4*>(y + n * 2) = pr(u);
        } else {
            float2 v2 = *reinterpret_cast<const float2*>(x + n * 2);
            if (mk[va] != 0) { v2.x = neg_inf(); v2.y = neg_inf(); *reinterpret_cast<float2*>(x + n * 2) = v2; }
            *reinterpret_cast<float2*>(y + n * 2) = make_float2(__builtin_amdgcn_exp2f(fmaf(v2.x, L2E, -m0)) * i0, __builtin_amdgcn_exp2f(fmaf(v2.y, L2E, -m1)) * i1);
        }
    }
}

__global__ __launch_bounds__(64) void tri_combine_kernel(const float* __restrict__ part, float* __restrict__ stats /* [B][G][2] */,
                                                         int G, int nchunk) {
    const int b = blockIdx.x, g = blockIdx.y, lane = threadIdx.x;
    const float* p = part + (int64_t)b * nchunk * G * 2;
    float m = neg_inf();
    for (int c = lane; c < nchunk; c += 64) m = fmaxf(m, p[((int64_t)c * G + g) * 2]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        const float mc = p[((int64_t)c * G + g) * 2], sc = p[((int64_t)c * G + g) * 2 + 1];
        if (mc != neg_inf()) s += sc * expf(mc - m);
    }
    s = wave_sum(s);
    if (lane == 0) {
        stats[((int64_t)b * G + g) * 2] = m;
        // an all-masked sample has m = -inf: the reference then computes exp(-inf - -inf) = NaN everywhere
        stats[((int64_t)b * G + g) * 2 + 1] = (m == neg_inf()) ? __builtin_nanf("") : s;
    }
}

__global__ __launch_bounds__(256) void tri_normalise_kernel(const float* __restrict__ logits, const float* __restrict__ stats,
                                                            float* __restrict__ p, int64_t NG, int G) {
    const int b = blockIdx.y;
    const float* x = logits + (int64_t)b * NG;
    float* y = p + (int64_t)b * NG;
    const float* st = stats + (int64_t)b * G * 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < NG; f += stride) {
        const int g = (int)(f % G);
        y[f] = expf(x[f] - st[g * 2]) / st[g * 2 + 1];
    }
}

// =====================================================================================================
// Bi softmax.  logits (B, G, N = V*Q): one wave per contiguous row                    (attention.py:35-39)
// =====================================================================================================
__global__ __launch_bounds__(256) void bi_softmax_kernel(float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                         float* __restrict__ p, int rows, int G, int V, int Q) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / G;
    const int N = V * Q;
    float* x = logits + (int64_t)row * N;
    float* y = p + (int64_t)row * N;
    const uint8_t* mk = mask ? mask + (int64_t)b * V : nullptr;
    float mx = neg_inf();
    for (int n = lane; n < N; n += 64) {
        float v = x[n];
        if (mk && mk[n / Q]) { v = neg_inf(); x[n] = v; }
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += expf(x[n] - mx);       // all-masked row: -inf - -inf = NaN, as torch
    s = wave_sum(s);
    for (int n = lane; n < N; n += 64) y[n] = expf(x[n] - mx) / s;
}

// =====================================================================================================
// tri pool: out[b,d] = sum_v vt[b,v,d] * sum_q qt[b,q,d] * sum_a w[b,v,q,a] * at[b,a,d]      (tc.py:59)
// lane axis = d (contiguous in vt/qt/at/out); w is workgroup-uniform.  at/qt slices for the workgroup's 256
// channels are staged in LDS when they fit, w[b] is read through the scalar/L1 path (uniform address).
// =====================================================================================================
__global__ __launch_bounds__(256) void tri_pool_kernel(const float* __restrict__ vt, const float* __restrict__ qt,
                                                       const float* __restrict__ at, const float* __restrict__ w,
                                                       int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa,
                                                       float* __restrict__ out, int V, int Q, int A, int D, int stage_a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y, t = threadIdx.x;
    const int d = blockIdx.x * 256 + t;
    const bool live = d < D;
    const int dd = live ? d : D - 1;
    float* qs = sm;                         // [Q][256]
    float* as = sm + (size_t)Q * 256;       // [A][256] when stage_a
    const float* qb = qt + (int64_t)b * Q * D;
    const float* ab = at + (int64_t)b * A * D;
    for (int q = 0; q < Q; ++q) qs[q * 256 + t] = qb[(int64_t)q * D + dd];
    if (stage_a) for (int a = 0; a < A; ++a) as[a * 256 + t] = ab[(int64_t)a * D + dd];
    // (each thread only ever reads back its own column: no barrier needed)
    const float* wb = w + (int64_t)b * w_sb;
    const float* vb = vt + (int64_t)b * V * D;
    float acc = 0.f;
    for (int v = 0; v < V; ++v) {
        float sv = 0.f;
        for (int q = 0; q < Q; ++q) {
            const float* wr = wb + v * w_sv + q * w_sq;
            float sq = 0.f;
            if (stage_a) { for (int a = 0; a < A; ++a) sq = fmaf(wr[a * w_sa], as[a * 256 + t], sq); }
            else         { for (int a = 0; a < A; ++a) sq = fmaf(wr[a * w_sa], ab[(int64_t)a * D + dd], sq); }
            sv = fmaf(sq, qs[q * 256 + t], sv);
        }
        acc = fmaf(sv, vb[(int64_t)v * D + dd], acc);
    }
    if (live) out[(int64_t)b * D + d] = acc;
}

// Small-A variant (A <= 8: the 3..6 answer tokens of the model configurations).  The attention slice w[b] (V*Q*A floats, strided
// in global memory: att[..., g]) is compacted into LDS once per workgroup as [v][q][AP] (AP = 4 or 8), so the inner loop is one
// broadcast ds_read_b128 per (v,q) instead of A dependent scalar loads; the thread's at[., d] column lives in registers.
template <int AP>
__global__ __launch_bounds__(256) void tri_pool_small_kernel(const float* __restrict__ vt, const float* __restrict__ qt,
                                                             const float* __restrict__ at, const float* __restrict__ w,
                                                             int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa,
                                                             float* __restrict__ out, int V, int Q, int A, int D) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y, t = threadIdx.x;
    const int d = blockIdx.x * 256 + t;
    const bool live = d < D;
    const int dd = live ? d : D - 1;
    float* wl = sm;                                     // [V*Q][AP]
    float* qs = sm + (size_t)V * Q * AP;                // [Q][256]
    const float* wb = w + (int64_t)b * w_sb;
    for (int i = t; i < V * Q * AP; i += 256) {
        const int a = i % AP, vq = i / AP, q = vq % Q, v = vq / Q;
        wl[i] = a < A ? wb[v * w_sv + q * w_sq + a * w_sa] : 0.f;
    }
    const float* qb = qt + (int64_t)b * Q * D;
    for (int q = 0; q < Q; ++q) qs[q * 256 + t] = qb[(int64_t)q * D + dd];
    float ar[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) ar[a] = a < A ? at[((int64_t)b * A + a) * D + dd] : 0.f;
    __syncthreads();
    const float* vb = vt + (int64_t)b * V * D;
    float acc = 0.f;
    for (int v = 0; v < V; ++v) {
        const float vv = vb[(int64_t)v * D + dd];
        float sv = 0.f;
        const float4* wr = reinterpret_cast<const float4*>(wl + (size_t)v * Q * AP);
        for (int q = 0; q < Q; ++q) {
            float sq;
            const float4 w0 = wr[q * (AP / 4)];
            sq = w0.x * ar[0] + w0.y * ar[1] + w0.z * ar[2] + w0.w * ar[3];
            if (AP == 8) {
                const float4 w1 = wr[q * 2 + 1];
                sq += w1.x * ar[4] + w1.y * ar[5] + w1.z * ar[6] + w1.w * ar[7];
            }
            sv = fmaf(sq, qs[q * 256 + t], sv);
        }
        acc = fmaf(sv, vv, acc);
    }
    if (live) out[(int64_t)b * D + d] = acc;
}

// =====================================================================================================
// bi pool: out[b,n] = sum_{j<k} sum_v vt[b,v,nk+j] * sum_q w[b,v,q] * qt[b,q,nk+j]        (bc.py:70-78)
// One thread per pooled output n; its k channels' qt column is staged in LDS ([q][j][lane], conflict-free).
// =====================================================================================================
__global__ __launch_bounds__(256) void bi_pool_kernel(const float* __restrict__ vt, const float* __restrict__ qt,
                                                      const float* __restrict__ w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                                                      float* __restrict__ out, int V, int Q, int D, int k) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [Q][k][256]
    const int b = blockIdx.y, t = threadIdx.x;
    const int NO = D / k;
    const int n = blockIdx.x * 256 + t;
    const bool live = n < NO;
    const int nn = live ? n : NO - 1;
    const float* qb = qt + (int64_t)b * Q * D;
    for (int q = 0; q < Q; ++q)
        for (int j = 0; j < k; ++j) sm[(q * k + j) * 256 + t] = qb[(int64_t)q * D + nn * k + j];
    const float* vb = vt + (int64_t)b * V * D;
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
        const int d = nn * k + j;
        if (w) {
            const float* wb = w + (int64_t)b * w_sb;
            for (int v = 0; v < V; ++v) {
                float sq = 0.f;
                for (int q = 0; q < Q; ++q) sq = fmaf(wb[v * w_sv + q * w_sq], sm[(q * k + j) * 256 + t], sq);
                acc = fmaf(sq, vb[(int64_t)v * D + d], acc);
            }
        } else {                                      // w == 1 (BCNet.forward, h_out=None)
            float sq = 0.f, sv = 0.f;
            for (int q = 0; q < Q; ++q) sq += sm[(q * k + j) * 256 + t];
            for (int v = 0; v < V; ++v) sv += vb[(int64_t)v * D + d];
            acc = fmaf(sq, sv, acc);
        }
    }
    if (live) out[(int64_t)b * NO + n] = acc;
}

// =====================================================================================================
// Streaming sum-pools (k = 1, D % 4 == 0, Q <= 16, A <= 8): the model configurations' K5 / K6.  These kernels are HBM-bound on
// the vt stream (B*V*D floats); what limits a naive loop is memory-level parallelism, not bandwidth.  Here a thread owns FOUR
// consecutive channels (16-B loads), a workgroup of 128 threads one 512-channel slab of one sample; the qt (and at) columns
// live in registers, the attention slice is compacted into LDS once, and the vt rows are fetched VC at a time -- VC independent
// 16-B loads in flight per thread (~24 KiB per workgroup, several workgroups per CU) -- before any of them is consumed.
// TRI: out[b,d] = sum_v vt[v,d] * sum_q qt[q,d] * sum_a w[v,q,a] at[a,d];   BI: out[b,d] = sum_v vt[v,d] * sum_q w[v,q] qt[q,d].
// =====================================================================================================
// The "shifted" form of the sum-pools (cti_*_pool_shift_fwd, inference): the q / a operand is relu(row + add[b, :]) formed as the rows are loaded --
// row = the hoisted pre-activation projection of the INITIAL sequence, add = the projection of the accumulated residual (one (B, D) vector per sample): the
// glimpse loops of src/FFOE/base_model.py:53-61,129-132 then need no (B*L, D) projection GEMM between two glimpses (base_model.py here, _HoistedLoop).
struct PoolShift { const float* qadd; const float* aadd; int relu; };
// Round 5 (the unrolled BAN glimpse loop, base_model.py `_ban_forward_unrolled`): the q operand's shift as a SUM of up to POOL_MAX_ADDS addends -- the raw split-K partials of
// the products that feed it, each (B, ld) fp32 with its own row stride (0 = one row broadcast over the batch) -- added up as the pool loads them: the reduce pass that
// stood between a product and the pool it feeds is gone.  n == 0: the single-addend form of PoolShift.
constexpr int POOL_MAX_ADDS = 32;
struct PoolAdds { const float* p[POOL_MAX_ADDS]; int ld[POOL_MAX_ADDS]; int n; int ldo; };      // ldo: row stride of `out` (0 = D): the pooled vectors of all glimpses side by side
__device__ __forceinline__ float shift1(float x, float a, int relu) { x += a; return relu ? fmaxf(x, 0.f) : x; }
// `c ? *p : z` with two lvalues is an lvalue conditional: the constant would live in private memory (scratch) and be loaded from there -- select VALUES instead
__device__ __forceinline__ float2 ld2_or_zero(bool c, const float* p) { float2 r = make_float2(0.f, 0.f); if (c) r = *reinterpret_cast<const float2*>(p); return r; }
__device__ __forceinline__ float4 ld4_or_zero(bool c, const float* p) { float4 r = make_float4(0.f, 0.f, 0.f, 0.f); if (c) r = *reinterpret_cast<const float4*>(p); return r; }
// Round 5: the projected `vt` rows may be bf16 (vt16 != 0: the plain-bf16 mode's hoisted projections write the pools' operand as bf16 rows -- half the bytes of
// the tensor these kernels exist to stream).  Four consecutive channels are one 16-B fp32 load or one 8-B bf16 load widened in registers.
typedef unsigned pl_u32x2 __attribute__((ext_vector_type(2)));

template <bool TRI, int AP, int NG, int QX>
__global__ __launch_bounds__(128 * NG) void pool_stream_kernel(const float* __restrict__ vt, const float* __restrict__ qt,
                                                          const float* __restrict__ at, const float* __restrict__ w,
                                                          int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa,
                                                          float* __restrict__ out, int V, int Q, int A, int D, PoolShift sh, int vt16, PoolAdds qa) {
#ifndef CTI_POOL_VC
#define CTI_POOL_VC 9
#endif
    // QX = the exact Q (12 or 14: no guards in the unrolled q loops, so the LDS reads of a whole v are issued together) or 0 = any Q <= 16
    constexpr int VC = CTI_POOL_VC, QM = 16, QL = QX ? QX : QM;
    const int Qn = QX ? QX : Q;
    extern __shared__ __attribute__((aligned(16))) float sm[];       // TRI: [V*Q][AP]   BI: [V][QM]
    // NG groups of 128 threads split the v range of the slab (more wavefronts in flight: the arithmetic of the tri pool is not
    // negligible -- V*Q*(A+1) FMAs per channel); their partial sums meet in LDS
    const int b = blockIdx.y, tt = threadIdx.x, t = tt & 127, grp = tt >> 7;
    const int d = (blockIdx.x * 128 + t) * 4;
    const bool live = d < D;
    const int dd = live ? d : 0;
    constexpr int NT = 128 * NG;
    if (w) {
        const float* wb = w + (int64_t)b * w_sb;
        if (TRI) {
            for (int i = tt; i < V * Q * AP; i += NT) {
                const int a = i % AP, vq = i / AP, q = vq % Q, v = vq / Q;
                sm[i] = a < A ? wb[v * w_sv + q * w_sq + a * w_sa] : 0.f;
            }
        } else {
            for (int i = tt; i < V * QM; i += NT) {
                const int q = i % QM, v = i / QM;
                sm[i] = q < Q ? wb[v * w_sv + q * w_sq] : 0.f;
            }
        }
    } else {
        for (int i = tt; i < V * QM; i += NT) sm[i] = (i % QM) < Q ? 1.f : 0.f;      // BCNet.forward with h_out = None: w == 1
    }
    float4 qr[QM];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < QM; ++q) qr[q] = ld4_or_zero(q < QL && q < Qn, qt + ((int64_t)b * Q + q) * D + dd);
    float4 ar[AP];
    if (TRI) {
#pragma unroll
        for (int a = 0; a < AP; ++a) ar[a] = ld4_or_zero(a < A, at + ((int64_t)b * A + a) * D + dd);
    }
    if (sh.relu || sh.qadd || sh.aadd || qa.n) {                     // shifted form: the real rows only (padding rows stay 0)
        float4 dq = ld4_or_zero(sh.qadd != nullptr, sh.qadd + (int64_t)b * D + dd);
        // several addends: the partial slabs of the products behind this shift, summed here.  Eight INDEPENDENT loads per round (the first form walked up to 31
        // addends one dependent load at a time: the pool took 26 us instead of 16.5, profiles/r05_model_c4_kernel_stats_unrolled_v1.txt)
        for (int i0 = 0; i0 < qa.n; i0 += 8) {
            float4 t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u < qa.n ? i0 + u : i0;           // (a short last round re-reads its first addend; its value is not added)
                t8[u] = *reinterpret_cast<const float4*>(qa.p[i] + (int64_t)b * qa.ld[i] + dd);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < qa.n) { dq.x += t8[u].x; dq.y += t8[u].y; dq.z += t8[u].z; dq.w += t8[u].w; }
        }
#pragma unroll
        for (int q = 0; q < QM; ++q)
            if (q < QL && q < Qn) { qr[q].x = shift1(qr[q].x, dq.x, sh.relu); qr[q].y = shift1(qr[q].y, dq.y, sh.relu); qr[q].z = shift1(qr[q].z, dq.z, sh.relu); qr[q].w = shift1(qr[q].w, dq.w, sh.relu); }
        if (TRI) {
            const float4 da = ld4_or_zero(sh.aadd != nullptr, sh.aadd + (int64_t)b * D + dd);
#pragma unroll
            for (int a = 0; a < AP; ++a)
                if (a < A) { ar[a].x = shift1(ar[a].x, da.x, sh.relu); ar[a].y = shift1(ar[a].y, da.y, sh.relu); ar[a].z = shift1(ar[a].z, da.z, sh.relu); ar[a].w = shift1(ar[a].w, da.w, sh.relu); }
        }
    }
    __syncthreads();
    const int vsz = vt16 ? 2 : 4;                                     // bytes per element of vt
    const char* vb = reinterpret_cast<const char*>(vt) + ((int64_t)b * V * D + dd) * vsz;
    float4 acc = z4;
    const int vper = (V + NG - 1) / NG, v_lo = grp * vper, v_hi = min(V, v_lo + vper);
    for (int v0 = v_lo; v0 < v_hi; v0 += VC) {
        float4 vr[VC];
#pragma unroll
        for (int u = 0; u < VC; ++u) vr[u] = z4;
        // the row format is decided ONCE around the whole round of loads (a per-load choice keeps the compiler from issuing them back to back)
        if (vt16) {
            pl_u32x2 raw[VC];
#pragma unroll
            for (int u = 0; u < VC; ++u) { raw[u] = pl_u32x2{0u, 0u}; if (v0 + u < v_hi) raw[u] = *reinterpret_cast<const pl_u32x2*>(vb + (int64_t)(v0 + u) * D * 2); }
#pragma unroll
            for (int u = 0; u < VC; ++u)
                vr[u] = make_float4(__builtin_bit_cast(float, raw[u][0] << 16), __builtin_bit_cast(float, raw[u][0] & 0xffff0000u),
                                    __builtin_bit_cast(float, raw[u][1] << 16), __builtin_bit_cast(float, raw[u][1] & 0xffff0000u));
        } else {
#pragma unroll
            for (int u = 0; u < VC; ++u) if (v0 + u < v_hi) vr[u] = *reinterpret_cast<const float4*>(vb + (int64_t)(v0 + u) * D * 4);
        }
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            const int v = v0 + u;
            if (v < v_hi) {
                float4 sv = z4;
                if (TRI) {
                    const float4* wr = reinterpret_cast<const float4*>(sm + (size_t)v * Q * AP);
#pragma unroll
                    for (int q = 0; q < QL; ++q) {
                        if (QX || q < Q) {
                            const float4 w0 = wr[q * (AP / 4)];
                            float4 sq;
                            sq.x = w0.x * ar[0].x + w0.y * ar[1].x + w0.z * ar[2].x + w0.w * ar[3].x;
                            sq.y = w0.x * ar[0].y + w0.y * ar[1].y + w0.z * ar[2].y + w0.w * ar[3].y;
                            sq.z = w0.x * ar[0].z + w0.y * ar[1].z + w0.z * ar[2].z + w0.w * ar[3].z;
                            sq.w = w0.x * ar[0].w + w0.y * ar[1].w + w0.z * ar[2].w + w0.w * ar[3].w;
                            if (AP == 8) {
                                const float4 w1 = wr[q * 2 + 1];
                                sq.x += w1.x * ar[AP - 4].x + w1.y * ar[AP - 3].x + w1.z * ar[AP - 2].x + w1.w * ar[AP - 1].x;
                                sq.y += w1.x * ar[AP - 4].y + w1.y * ar[AP - 3].y + w1.z * ar[AP - 2].y + w1.w * ar[AP - 1].y;
                                sq.z += w1.x * ar[AP - 4].z + w1.y * ar[AP - 3].z + w1.z * ar[AP - 2].z + w1.w * ar[AP - 1].z;
                                sq.w += w1.x * ar[AP - 4].w + w1.y * ar[AP - 3].w + w1.z * ar[AP - 2].w + w1.w * ar[AP - 1].w;
                            }
                            sv.x = fmaf(sq.x, qr[q].x, sv.x); sv.y = fmaf(sq.y, qr[q].y, sv.y);
                            sv.z = fmaf(sq.z, qr[q].z, sv.z); sv.w = fmaf(sq.w, qr[q].w, sv.w);
                        }
                    }
                } else {
                    const float4* wr = reinterpret_cast<const float4*>(sm + (size_t)v * QM);
#pragma unroll
                    for (int q4 = 0; q4 < (QL + 3) / 4; ++q4) {
                        if (QX || q4 * 4 < Q) {
                            const float4 ww = wr[q4];
                            sv.x += ww.x * qr[q4 * 4].x + ww.y * qr[q4 * 4 + 1].x + ww.z * qr[q4 * 4 + 2].x + ww.w * qr[q4 * 4 + 3].x;
                            sv.y += ww.x * qr[q4 * 4].y + ww.y * qr[q4 * 4 + 1].y + ww.z * qr[q4 * 4 + 2].y + ww.w * qr[q4 * 4 + 3].y;
                            sv.z += ww.x * qr[q4 * 4].z + ww.y * qr[q4 * 4 + 1].z + ww.z * qr[q4 * 4 + 2].z + ww.w * qr[q4 * 4 + 3].z;
                            sv.w += ww.x * qr[q4 * 4].w + ww.y * qr[q4 * 4 + 1].w + ww.z * qr[q4 * 4 + 2].w + ww.w * qr[q4 * 4 + 3].w;
                        }
                    }
                }
                acc.x = fmaf(sv.x, vr[u].x, acc.x); acc.y = fmaf(sv.y, vr[u].y, acc.y);
                acc.z = fmaf(sv.z, vr[u].z, acc.z); acc.w = fmaf(sv.w, vr[u].w, acc.w);
            }
        }
    }
    if (NG > 1) {
        __syncthreads();                                               // the attention slice is dead: reuse LDS for the partials
        float4* red = reinterpret_cast<float4*>(sm);
        if (grp > 0) red[(grp - 1) * 128 + t] = acc;
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < NG; ++g) { const float4 o = red[(g - 1) * 128 + t]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        }
    }
    if (live && grp == 0) *reinterpret_cast<float4*>(out + (int64_t)b * (qa.ldo ? qa.ldo : D) + d) = acc;
}

// Tri pool, product-table form (exact Q*A known at compile time: 36, 42 -- the A = 3 answer tokens of the FFOE model).  The arithmetic of the tri pool is V*Q*A FMAs per
// channel -- 13.6 flop per HBM byte, near the fp32 VALU ridge -- so the instruction count decides.  A thread owns TWO channels and
// first forms P[q,a] = qt[q,d] * at[a,d] (Q*A float2 registers); every v is then ONE dot product of the compacted attention row
// w[v, 0:QA] (LDS, broadcast ds_read_b128) with P: QA packed FMAs per v and channel pair, no padding of A, no per-q re-association.
#ifndef CTI_TP_ABL
#define CTI_TP_ABL 0            // timing-only ablations (wrong results): 1 one packed-FMA group per object instead of QA/4, 2 no LDS reads of the attention row,
#endif                          // 4 only the first v group is loaded, 8 no attention-slice compaction.  Measured (profiles/r03_tri_pool_ablation.txt): 24.9 us; 1: 14.0;
                                // 2: 21.9; 4: 22.9; 8: 20.0; all: 10.2 -- a one-round kernel whose phases add up on a 10-us floor, not one saturated resource
template <int QA, int NG, int AC>
__global__ __launch_bounds__(128 * NG) void tri_pool_table_kernel(const float* __restrict__ vt, const float* __restrict__ qt,
                                                                  const float* __restrict__ at, const float* __restrict__ w,
                                                                  int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa,
                                                                  float* __restrict__ out, int V, int Q, int A, int D, PoolShift sh) {
    constexpr int QAP = (QA + 3) & ~3, VC = 9;
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [V][QAP]
    const int b = blockIdx.y, tt = threadIdx.x, t = tt & 127, grp = tt >> 7;
    const int d = (blockIdx.x * 128 + t) * 2;
    const bool live = d < D;
    const int dd = live ? d : 0;
    constexpr int NT = 128 * NG;
    // Round 3: the kernel ran as a CHAIN of exposed global-load latencies -- the attention slice, then the q / a rows, then four groups of
    // nine v rows, each waited for before the next was issued, with two waves per SIMD to hide them (the whole grid is one round of 1 024
    // two-wave workgroups).  Now every independent load is in flight before the first wait: the first v group and the q / a rows are
    // issued ahead of the attention slice's compaction, and v group c + 1 is loaded under group c's arithmetic.
    const float* vb = vt + (int64_t)b * V * D + dd;
    const int vper = (V + NG - 1) / NG, v_lo = grp * vper, v_hi = min(V, v_lo + vper);
    float2 vr[VC], vn[VC];
#pragma unroll
    for (int u = 0; u < VC; ++u) vr[u] = v_lo + u < v_hi ? *reinterpret_cast<const float2*>(vb + (int64_t)(v_lo + u) * D) : make_float2(0.f, 0.f);
    constexpr int QC = QA / AC;                                      // Q and A are compile-time here (A == AC, Q == QC: the launcher checks)
    float2 ar[AC], qr[QC];
#pragma unroll
    for (int a = 0; a < AC; ++a) ar[a] = *reinterpret_cast<const float2*>(at + ((int64_t)b * AC + a) * D + dd);
#pragma unroll
    for (int q = 0; q < QC; ++q) qr[q] = *reinterpret_cast<const float2*>(qt + ((int64_t)b * QC + q) * D + dd);
    if (sh.relu || sh.qadd || sh.aadd) {
        const float2 dq = sh.qadd ? *reinterpret_cast<const float2*>(sh.qadd + (int64_t)b * D + dd) : make_float2(0.f, 0.f);
        const float2 da = sh.aadd ? *reinterpret_cast<const float2*>(sh.aadd + (int64_t)b * D + dd) : make_float2(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < QC; ++q) { qr[q].x = shift1(qr[q].x, dq.x, sh.relu); qr[q].y = shift1(qr[q].y, dq.y, sh.relu); }
#pragma unroll
        for (int a = 0; a < AC; ++a) { ar[a].x = shift1(ar[a].x, da.x, sh.relu); ar[a].y = shift1(ar[a].y, da.y, sh.relu); }
    }
    const float* wb = w + (int64_t)b * w_sb;
    if (!(CTI_TP_ABL & 8)) {
    for (int i = tt; i < V * QAP; i += NT) {
        const int qa = i % QAP, v = i / QAP, q = qa / AC, a = qa - q * AC;      // (compile-time divisors)
        sm[i] = qa < QA ? wb[v * w_sv + q * w_sq + a * w_sa] : 0.f;
    }
    }
    float2 P[QAP];
#pragma unroll
    for (int i = 0; i < QAP; ++i)
        P[i] = i < QA ? make_float2(qr[i / AC].x * ar[i % AC].x, qr[i / AC].y * ar[i % AC].y) : make_float2(0.f, 0.f);
    __syncthreads();
    float2 acc = make_float2(0.f, 0.f);
    for (int v0 = v_lo; v0 < v_hi; v0 += VC) {
#pragma unroll
        for (int u = 0; u < VC; ++u) vn[u] = (v0 + VC + u < v_hi && !(CTI_TP_ABL & 4)) ? *reinterpret_cast<const float2*>(vb + (int64_t)(v0 + VC + u) * D) : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            const int v = v0 + u;
            if (v < v_hi) {
                const float4* wr = reinterpret_cast<const float4*>(sm + (size_t)v * QAP);
                float2 s0 = make_float2(0.f, 0.f), s1 = s0;
#pragma unroll
                for (int i4 = 0; i4 < ((CTI_TP_ABL & 1) ? 1 : QAP / 4); ++i4) {
                    const float4 ww = (CTI_TP_ABL & 2) ? make_float4(vr[u].x, vr[u].y, vr[u].x, vr[u].y) : wr[i4];
                    s0.x = fmaf(ww.x, P[i4 * 4].x, s0.x);     s0.y = fmaf(ww.x, P[i4 * 4].y, s0.y);
                    s1.x = fmaf(ww.y, P[i4 * 4 + 1].x, s1.x); s1.y = fmaf(ww.y, P[i4 * 4 + 1].y, s1.y);
                    s0.x = fmaf(ww.z, P[i4 * 4 + 2].x, s0.x); s0.y = fmaf(ww.z, P[i4 * 4 + 2].y, s0.y);
                    s1.x = fmaf(ww.w, P[i4 * 4 + 3].x, s1.x); s1.y = fmaf(ww.w, P[i4 * 4 + 3].y, s1.y);
                }
                acc.x = fmaf(s0.x + s1.x, vr[u].x, acc.x);
                acc.y = fmaf(s0.y + s1.y, vr[u].y, acc.y);

            }
        }
#pragma unroll
        for (int u = 0; u < VC; ++u) vr[u] = vn[u];
    }
    if (NG > 1) {
        __syncthreads();
        float2* red = reinterpret_cast<float2*>(sm);
        if (grp > 0) red[(grp - 1) * 128 + t] = acc;
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < NG; ++g) { const float2 o = red[(g - 1) * 128 + t]; acc.x += o.x; acc.y += o.y; }
        }
    }
    if (live && grp == 0) *reinterpret_cast<float2*>(out + (int64_t)b * D + d) = acc;
}

// Bi pool with k = 3 (BCNet(k=3).forward_with_weights, reference src/bc.py:73-77: the matmul pair over the 3*h channels, then AvgPool1d(3) * 3 = the
// sum of each group of three).  A thread owns SIX consecutive channels = two output columns, so the pooling never crosses threads: three 8-B loads
// per row, qt columns in registers, the attention slice compacted into LDS, v groups loaded one group ahead of their arithmetic (as in the tri pool's
// table kernel).  Replaces the generic bi_pool_kernel at this shape: 193 us -> measured in profiles/r03_hbm_kernels.jsonl (B = 256, V = 36, Q = 14, D = 3072).
template <int NG>
__global__ __launch_bounds__(128 * NG, 2) void bi_pool_k3_kernel(const float* __restrict__ vt, const float* __restrict__ qt, const float* __restrict__ w,
                                                              int64_t w_sb, int64_t w_sv, int64_t w_sq, float* __restrict__ out, int V, int Q, int D) {
    constexpr int QM = 16, VC = 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [V][QM]
    const int b = blockIdx.y, tt = threadIdx.x, t = tt & 127, grp = tt >> 7;
    const int d = (blockIdx.x * 128 + t) * 6;
    const bool live = d < D;
    const int dd = live ? d : 0;
    constexpr int NT = 128 * NG;
    const float* vb = vt + (int64_t)b * V * D + dd;
    const int vper = (V + NG - 1) / NG, v_lo = grp * vper, v_hi = min(V, v_lo + vper);
    const float2 z2 = make_float2(0.f, 0.f);
    float2 vr[VC][3], vn[VC][3];
#pragma unroll
    for (int u = 0; u < VC; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c) vr[u][c] = ld2_or_zero(v_lo + u < v_hi, vb + (int64_t)(v_lo + u) * D + 2 * c);
    float2 qr[QM][3];
#pragma unroll
    for (int q = 0; q < QM; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) qr[q][c] = ld2_or_zero(q < Q, qt + ((int64_t)b * Q + q) * D + dd + 2 * c);
    if (w) {
        const float* wb = w + (int64_t)b * w_sb;
        for (int i = tt; i < V * QM; i += NT) { const int q = i % QM, v = i / QM; sm[i] = q < Q ? wb[v * w_sv + q * w_sq] : 0.f; }
    } else {
        for (int i = tt; i < V * QM; i += NT) sm[i] = (i % QM) < Q ? 1.f : 0.f;
    }
    __syncthreads();
    float2 acc[3] = {z2, z2, z2};
    for (int v0 = v_lo; v0 < v_hi; v0 += VC) {
#pragma unroll
        for (int u = 0; u < VC; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) vn[u][c] = ld2_or_zero(v0 + VC + u < v_hi, vb + (int64_t)(v0 + VC + u) * D + 2 * c);
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            const int v = v0 + u;
            if (v < v_hi) {
                const float4* wr = reinterpret_cast<const float4*>(sm + (size_t)v * QM);
                float2 sv[3] = {z2, z2, z2};
#pragma unroll
                for (int q4 = 0; q4 < QM / 4; ++q4) {
                    const float4 ww = wr[q4];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        sv[c].x += ww.x * qr[q4 * 4][c].x + ww.y * qr[q4 * 4 + 1][c].x + ww.z * qr[q4 * 4 + 2][c].x + ww.w * qr[q4 * 4 + 3][c].x;
                        sv[c].y += ww.x * qr[q4 * 4][c].y + ww.y * qr[q4 * 4 + 1][c].y + ww.z * qr[q4 * 4 + 2][c].y + ww.w * qr[q4 * 4 + 3][c].y;
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) { acc[c].x = fmaf(sv[c].x, vr[u][c].x, acc[c].x); acc[c].y = fmaf(sv[c].y, vr[u][c].y, acc[c].y); }
            }
        }
#pragma unroll
        for (int u = 0; u < VC; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) vr[u][c] = vn[u][c];
    }
    // channels d .. d+5 = (acc[0].x, acc[0].y, acc[1].x | acc[1].y, acc[2].x, acc[2].y): the two pooled outputs
    float2 o = make_float2(acc[0].x + acc[0].y + acc[1].x, acc[1].y + acc[2].x + acc[2].y);
    if (NG > 1) {
        __syncthreads();
        float2* red = reinterpret_cast<float2*>(sm);
        if (grp > 0) red[(grp - 1) * 128 + t] = o;
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int g = 1; g < NG; ++g) { const float2 r = red[(g - 1) * 128 + t]; o.x += r.x; o.y += r.y; }
        }
    }
    if (live && grp == 0) *reinterpret_cast<float2*>(out + (int64_t)b * (D / 3) + d / 3) = o;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// =====================================================================================================
// bi logits: logits[b,g,v,q] = hs * sum_d vt[b,v,d] h[g,d] qt[b,q,d] + hb[g]               (bc.py:52-58)
// ONE WAVE per (b, v): the 64 lanes split the d axis (coalesced reads of the vt row, of h and of the qt rows, the
// latter two L2-resident), each lane keeps a GC x QC block of partial sums, reduced by wave shuffles.  A wave per
// row keeps the ratio of shuffle-reduction work to FMAs at 1:4 (with 256 threads per row it was 1:1).
// =====================================================================================================
constexpr int GC = 8, QC = 16;
__global__ __launch_bounds__(64) void bi_logits_kernel(const float* __restrict__ vt, const float* __restrict__ qt,
                                                       const float* __restrict__ h, const float* __restrict__ h_scale,
                                                       const float* __restrict__ h_bias, float* __restrict__ logits,
                                                       int G, int V, int Q, int D) {
    const int bv = blockIdx.x, b = bv / V, v = bv % V, lane = threadIdx.x;
    const float* vrow = vt + (int64_t)bv * D;
    const float* qb = qt + (int64_t)b * Q * D;
    const float hs = h_scale ? h_scale[0] : 1.f;
    for (int g0 = 0; g0 < G; g0 += GC) {
        for (int q0 = 0; q0 < Q; q0 += QC) {
            float acc[GC][QC];
#pragma unroll
            for (int g = 0; g < GC; ++g)
#pragma unroll
                for (int q = 0; q < QC; ++q) acc[g][q] = 0.f;
            for (int d = lane; d < D; d += 64) {
                const float x = vrow[d];
                float qv[QC];
#pragma unroll
                for (int q = 0; q < QC; ++q) qv[q] = (q0 + q < Q) ? qb[(int64_t)(q0 + q) * D + d] : 0.f;
#pragma unroll
                for (int g = 0; g < GC; ++g) {
                    const float xh = (g0 + g < G) ? x * h[(int64_t)(g0 + g) * D + d] : 0.f;
#pragma unroll
                    for (int q = 0; q < QC; ++q) acc[g][q] = fmaf(xh, qv[q], acc[g][q]);
                }
            }
#pragma unroll
            for (int g = 0; g < GC; ++g)
#pragma unroll
                for (int q = 0; q < QC; ++q) {
                    const float s = wave_sum(acc[g][q]);
                    if (lane == 0 && g0 + g < G && q0 + q < Q) {
                        const float hb = h_bias ? h_bias[g0 + g] : 0.f;
                        logits[(((int64_t)b * G + g0 + g) * V + v) * Q + q0 + q] = s * hs + hb;
                    }
                }
        }
    }
}

// =====================================================================================================
// bi logits on the MFMA (fp32-grade, 3 bf16 products): per sample a (V x D) . (G*Q x D)^T contraction whose right operand
// h[g,d] * qt[b,q,d] is formed on the fly.  Both operands have the contraction axis d contiguous in memory, and an MFMA
// 32x32x16 fragment is "8 consecutive k of one row" per lane -- so every lane loads its fragments STRAIGHT from global
// memory as two 16-B loads (no LDS, no layout pass), multiplies / splits them into bf16 hi + lo in registers and issues
// al*bh + ah*bl + ah*bh.  One 512-thread workgroup per (sample, half of D): its 8 waves own the (m-tile, n-tile) pairs;
// the two D halves meet by atomicAdd (two addends: order-independent).  716 us -> ~60 us at B=256, G=8, D=3072.
// =====================================================================================================
// zero fill as a KERNEL: a hipMemsetAsync captured into a hipGraph did not re-zero the buffer on the second replay (observed on ROCm 7.0:
// the accumulating kernels below then added onto stale data), a kernel node does
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
static inline int zero_fill(float* p, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
    return launch_status("zero_fill");
}

typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));
typedef float lf32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split8(const float4 a, const float4 b, lbf16x8& hi, lbf16x8& lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = static_cast<__bf16>(x[e]);
        hi[e] = h;
        lo[e] = static_cast<__bf16>(x[e] - static_cast<float>(h));
    }
}

__global__ __launch_bounds__(512) void bi_logits_mfma_kernel(const float* __restrict__ vt, const float* __restrict__ qt,
                                                             const float* __restrict__ h, const float* __restrict__ h_scale,
                                                             const float* __restrict__ h_bias, float* __restrict__ logits,
                                                             int G, int V, int Q, int D, int MT, int NT, int dper) {
    const int b = blockIdx.x, ks = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, kg = lane >> 5;
    const int N = G * Q;
    const int d_lo = ks * dper, d_hi = min(D, d_lo + dper);
    const float hs = h_scale ? h_scale[0] : 1.f;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tile = wid; tile < MT * NT; tile += 8) {
        const int mt = tile % MT, nt = tile / MT;
        const int v = mt * 32 + r, c = nt * 32 + r;
        const bool vok = v < V, cok = c < N;
        const int g = cok ? c / Q : 0, q = cok ? c - g * Q : 0;
        const float* ap = vt + ((int64_t)b * V + (vok ? v : 0)) * D + kg * 8;
        const float* hp = h + (int64_t)g * D + kg * 8;
        const float* qp = qt + ((int64_t)b * Q + q) * D + kg * 8;
        lf32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        float4 a0 = z4, a1 = z4, h0 = z4, h1 = z4, q0 = z4, q1 = z4;
        if (d_lo < d_hi) {
            if (vok) { a0 = *reinterpret_cast<const float4*>(ap + d_lo); a1 = *reinterpret_cast<const float4*>(ap + d_lo + 4); }
            if (cok) { h0 = *reinterpret_cast<const float4*>(hp + d_lo); h1 = *reinterpret_cast<const float4*>(hp + d_lo + 4);
                       q0 = *reinterpret_cast<const float4*>(qp + d_lo); q1 = *reinterpret_cast<const float4*>(qp + d_lo + 4); }
        }
        for (int d0 = d_lo; d0 < d_hi; d0 += 16) {
            const float4 ca0 = a0, ca1 = a1;
            const float4 p0 = make_float4(h0.x * q0.x, h0.y * q0.y, h0.z * q0.z, h0.w * q0.w);
            const float4 p1 = make_float4(h1.x * q1.x, h1.y * q1.y, h1.z * q1.z, h1.w * q1.w);
            const int dn = d0 + 16;
            if (dn < d_hi) {                                                   // next slice's raw fragments fly under this slice's arithmetic
                if (vok) { a0 = *reinterpret_cast<const float4*>(ap + dn); a1 = *reinterpret_cast<const float4*>(ap + dn + 4); }
                if (cok) { h0 = *reinterpret_cast<const float4*>(hp + dn); h1 = *reinterpret_cast<const float4*>(hp + dn + 4);
                           q0 = *reinterpret_cast<const float4*>(qp + dn); q1 = *reinterpret_cast<const float4*>(qp + dn + 4); }
            }
            lbf16x8 ah, al, bh, bl;
            split8(ca0, ca1, ah, al);
            split8(p0, p1, bh, bl);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
        if (cok) {
            const float hb = (h_bias && ks == 0) ? h_bias[g] : 0.f;
            float* o = logits + (((int64_t)b * G + g) * V) * Q + q;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int vv = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
                if (vv < V) atomicAdd(o + (int64_t)vv * Q, acc[e] * hs + hb);
            }
        }
    }
}

// The same contraction with the LEFT operand staged once per workgroup: V <= 64, G*Q <= 128, D % 32 == 0.  In the kernel above every wave
// splits its own fragments of vt and of h*qt into bf16 hi + lo -- ~90 VALU instructions per 3 MFMAs, with vt split again by every wave that
// shares its rows -- and the VALU, not the MFMA or HBM, sets its time (182 us at B = 256, G = 8, D = 3072).  Here a 32-deep slice of vt is split
// ONCE per workgroup into LDS (double-buffered, one barrier per slice); a wave owns two 16-column tiles of (g, q) -- its h*qt fragments are
// formed and split once and meet all V/16 row tiles -- on the 16x16x32 MFMA (36 rows pad to 48 instead of 64, 112 columns are 7 tiles exactly).
typedef float lf32x4 __attribute__((ext_vector_type(4)));
template <int TERMS>
__device__ __forceinline__ void split8t(const float4 a, const float4 b, lbf16x8& hi, lbf16x8& lo) {
    if (TERMS == 3) { split8(a, b, hi, lo); return; }
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) hi[e] = static_cast<__bf16>(x[e]);
    lo = hi;                                                       // (unused)
}

// TERMS = 1 (round 4): the plain-bf16 mode's form -- one product per pair, no lo parts: half the split work (the kernel's bound) and a third of the MFMAs
template <int TERMS>
__global__ __launch_bounds__(256) void bi_logits_lds_kernel(const float* __restrict__ vt, const float* __restrict__ qt, const float* __restrict__ h,
                                                            const float* __restrict__ h_scale, const float* __restrict__ h_bias,
                                                            float* __restrict__ logits, int G, int V, int Q, int D, int MT, int NT, int dper, int atomic, int NTW,
                                                            const uint8_t* __restrict__ sm_mask, float* __restrict__ sm_p, int* __restrict__ sm_cnt, int sm_parts, int vt16) {
    __shared__ __attribute__((aligned(16))) unsigned short As[2][2][64][40];      // [buffer][hi | lo][row v][32 k + 8 pad]
    const int b = blockIdx.x, ks = blockIdx.y;
    // blockIdx.z: which group of NTW column tiles this workgroup owns (round-3 experiment: thinner workgroups to put more loads in flight -- measured
    // SLOWER, see the launcher: the default keeps all tiles in one workgroup).
    const int ct0 = blockIdx.z * NTW;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int N = G * Q;
    const int d_lo = ks * dper, d_hi = min(D, d_lo + dper);
    const float hs = h_scale ? h_scale[0] : 1.f;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // staging role: thread -> (row, 8 consecutive k) of the vt slice
    const int sr = t >> 2, skq = t & 3;
    const bool sok = sr < V;
    // (round 5) vt16: the rows are bf16 already -- a thread's eight values are ONE 16-B load that goes to the hi image as it is (lo = 0)
    const int vsz = vt16 ? 2 : 4;
    const char* sp = reinterpret_cast<const char*>(vt) + (((int64_t)b * V + (sok ? sr : 0)) * D + skq * 8) * vsz;
    auto stage_load = [&](int d, float4& r0, float4& r1) {
        if (vt16) r0 = *reinterpret_cast<const float4*>(sp + (int64_t)d * 2);
        else { r0 = *reinterpret_cast<const float4*>(sp + (int64_t)d * 4); r1 = *reinterpret_cast<const float4*>(sp + (int64_t)d * 4 + 16); }
    };
    auto stage_split = [&](const float4 r0, const float4 r1, lbf16x8& hi, lbf16x8& lo) {
        if (vt16) {
            hi = __builtin_bit_cast(lbf16x8, r0);
#pragma unroll
            for (int e = 0; e < 8; ++e) lo[e] = static_cast<__bf16>(0.f);
        } else split8t<TERMS>(r0, r1, hi, lo);
    };
    // right-operand role: this wave's column tiles wid and wid + 4
    bool cok[2]; int cg[2], cq[2];
    const float* hp[2]; const float* qp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = (ct0 + wid + 4 * j) * 16 + l15;
        cok[j] = (wid + 4 * j) < NTW && (ct0 + wid + 4 * j) < NT && c < N;
        cg[j] = cok[j] ? c / Q : 0; cq[j] = cok[j] ? c - cg[j] * Q : 0;
        hp[j] = h + (int64_t)cg[j] * D + kq * 8;
        qp[j] = qt + ((int64_t)b * Q + cq[j]) * D + kq * 8;
    }
    lf32x4 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[j][m] = lf32x4{0.f, 0.f, 0.f, 0.f};
    {
        float4 a0 = z4, a1 = z4;
        if (sok && d_lo < d_hi) stage_load(d_lo, a0, a1);
        lbf16x8 hi, lo;
        stage_split(a0, a1, hi, lo);
        *reinterpret_cast<lbf16x8*>(&As[0][0][sr][skq * 8]) = hi;
        if (TERMS == 3) *reinterpret_cast<lbf16x8*>(&As[0][1][sr][skq * 8]) = lo;
    }
    __syncthreads();
    float4 rh0[2], rh1[2], rq0[2], rq1[2];                           // raw h / qt fragments of the CURRENT slice (loaded one slice ahead)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        rh0[j] = rh1[j] = rq0[j] = rq1[j] = z4;
        if (cok[j] && d_lo < d_hi) {
            rh0[j] = *reinterpret_cast<const float4*>(hp[j] + d_lo); rh1[j] = *reinterpret_cast<const float4*>(hp[j] + d_lo + 4);
            rq0[j] = *reinterpret_cast<const float4*>(qp[j] + d_lo); rq1[j] = *reinterpret_cast<const float4*>(qp[j] + d_lo + 4);
        }
    }
    int buf = 0;
    for (int d0 = d_lo; d0 < d_hi; d0 += 32, buf ^= 1) {
        const bool more = d0 + 32 < d_hi;
        lbf16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (wid + 4 * j < NTW && ct0 + wid + 4 * j < NT)        // uniform
                split8t<TERMS>(make_float4(rh0[j].x * rq0[j].x, rh0[j].y * rq0[j].y, rh0[j].z * rq0[j].z, rh0[j].w * rq0[j].w),
                               make_float4(rh1[j].x * rq1[j].x, rh1[j].y * rq1[j].y, rh1[j].z * rq1[j].z, rh1[j].w * rq1[j].w), bh[j], bl[j]);
        float4 n0 = z4, n1 = z4;
        if (more) {                                                  // the next slice's loads fly under this slice's MFMAs
            if (sok) stage_load(d0 + 32, n0, n1);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (cok[j]) {
                    rh0[j] = *reinterpret_cast<const float4*>(hp[j] + d0 + 32); rh1[j] = *reinterpret_cast<const float4*>(hp[j] + d0 + 36);
                    rq0[j] = *reinterpret_cast<const float4*>(qp[j] + d0 + 32); rq1[j] = *reinterpret_cast<const float4*>(qp[j] + d0 + 36);
                }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < MT) {                                            // uniform
                const lbf16x8 ah = *reinterpret_cast<const lbf16x8*>(&As[buf][0][m * 16 + l15][kq * 8]);
                lbf16x8 al = ah;
                if (TERMS == 3) al = *reinterpret_cast<const lbf16x8*>(&As[buf][1][m * 16 + l15][kq * 8]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (wid + 4 * j < NTW && ct0 + wid + 4 * j < NT) {
                        if (TERMS == 3) {
                            acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[j][m], 0, 0, 0);
                            acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[j][m], 0, 0, 0);
                        }
                        acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[j][m], 0, 0, 0);
                    }
                }
            }
        }
        if (more) {
            lbf16x8 hi, lo;
            stage_split(n0, n1, hi, lo);
            *reinterpret_cast<lbf16x8*>(&As[buf ^ 1][0][sr][skq * 8]) = hi;
            if (TERMS == 3) *reinterpret_cast<lbf16x8*>(&As[buf ^ 1][1][sr][skq * 8]) = lo;
        }
        __syncthreads();                                             // the next slice is complete; everyone is done reading this one
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (cok[j]) {
            const float hb = (h_bias && ks == 0) ? h_bias[cg[j]] : 0.f;
            float* o = logits + (((int64_t)b * G + cg[j]) * V) * Q + cq[j];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {                        // C/D: column = lane & 15, row = 4 * (lane >> 4) + i
                    const int v = m * 16 + 4 * kq + i;
                    if (m < MT && v < V) {
                        const float val = acc[j][m][i] * hs + hb;
                        if (atomic) atomicAdd(o + (int64_t)v * Q, val); else o[(int64_t)v * Q] = val;
                    }
                }
        }
    }
    // ---- BiAttention's mask + softmax in the same launch (round 3; reference src/attention.py:35-39).  A sample's G*V*Q logits (4 032 at the FFOE
    // shape) are complete once the sm_parts workgroups that add into them have passed this point: the LAST of them (a per-sample counter that it
    // resets for the next call) reads them back from the L2, fills the masked rows with -inf and writes p -- a wave per glimpse, V*Q values over
    // its lanes.  Replaces a 10-us launch over 8 MB.
    if (sm_p == nullptr) return;
    __shared__ int sm_last;
    __threadfence();
    __syncthreads();
    if (t == 0) {
        sm_last = 1;
        if (sm_parts > 1) {
            sm_last = atomicAdd(sm_cnt + b, 1) == sm_parts - 1;
            if (sm_last) atomicExch(sm_cnt + b, 0);
        }
    }
    __syncthreads();
    if (!sm_last) return;
    __threadfence();
    const int n = V * Q;
    const float ninf = -__builtin_huge_valf();
    for (int g = wid; g < G; g += 4) {
        float* lg = logits + ((int64_t)b * G + g) * n;
        float* pg = sm_p + ((int64_t)b * G + g) * n;
        float x[16];                                                     // n <= 64 * 16 (V <= 64, Q <= 16: the launcher checks)
        float mx = ninf;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = lane + 64 * i;
            x[i] = ninf;
            if (e < n) {
                const float val = lg[e];       // plain loads behind the acquire fence above (it invalidates this CU's L1): sixteen in flight at once -- as agent-scope atomic loads they were
                                               // issued one L2 round trip at a time, 37 us of tail
                const bool masked = sm_mask != nullptr && sm_mask[(int64_t)b * V + e / Q] != 0;
                x[i] = masked ? ninf : val;
                if (masked) lg[e] = ninf;
            }
            mx = fmaxf(mx, x[i]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int e = lane + 64 * i; if (e < n) { x[i] = __expf(x[i] - mx); sum += x[i]; } }     // all masked: -inf - -inf = NaN, like the reference
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int e = lane + 64 * i; if (e < n) pg[e] = x[i] * inv; }
    }
}

// =====================================================================================================
// Round 6: the same contraction WITHOUT a barrier in its K loop.  The LDS form above walks K in 32-deep slices behind a workgroup
// barrier each, with one slice of loads in flight: 48 dependent rounds of ~1.6 us = 77 us at B = 256, G = 8, D = 3 072 for 105 MB and 6 GFLOP.
// Here ONE workgroup of eight waves takes a sample and every wave a contiguous EIGHTH of K, for which it computes the whole (3 row tiles of objects)
// x (G column tiles) output on its own; the eight partial outputs meet in LDS at the end in a fixed order (deterministic; no atomics, no zero fill).
// Column tile g = glimpse g's Q <= 16 question positions (two dummy columns at Q = 14), so that
//   * the qt fragment of a K step (row q = lane & 15, eight k) is loaded ONCE and serves all G tiles,
//   * h[g, k] is the same for the 16 lanes of a k group: the wave copies its own K range of h into its own part of LDS once (12 KiB; wave-private, no
//     barrier) and reads 32 B per tile and K step from there (broadcast reads),
//   * per K step a wave issues 5 (bf16 vt rows) or 8 (fp32 rows) 1-KiB global loads for 3 G MFMAs -- the LDS form issued 8 per 8 MFMAs and fetched every
//     qt value eight times -- with the next step's operands in flight in a second register set.
// The rows' operands are converted exactly as before (fp32 product h * qt rounded to bf16 once; fp32-grade mode: hi + lo of both sides, three products).
template <int TERMS, int VT16>
__global__ __launch_bounds__(512) void bi_logits_ks_kernel(const void* __restrict__ vt_, const float* __restrict__ qt, const float* __restrict__ h,
                                                           const float* __restrict__ h_scale, const float* __restrict__ h_bias,
                                                           float* __restrict__ logits, int G, int V, int Q, int D, int spw) {
    extern __shared__ __attribute__((aligned(16))) float ksm[];     // [8 waves][G][spw * 32] of h, then the partial outputs
    constexpr int MT = 3, GT = 8;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int nsteps = D / 32, s0 = wid * spw, my = max(0, min(nsteps, s0 + spw) - s0);
    const int hw = spw * 32;                                         // floats per glimpse in a wave's slice
    float* hl = ksm + (size_t)wid * G * hw;
    for (int i = lane; i < G * my * 8; i += 64) {                    // (glimpse, float4 of this wave's K range)
        const int g = i / (my * 8), c = i - g * (my * 8);
        *reinterpret_cast<float4*>(hl + g * hw + c * 4) = *reinterpret_cast<const float4*>(h + (int64_t)g * D + s0 * 32 + c * 4);
    }
    constexpr int vsz = VT16 ? 2 : 4;
    const char* ap[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ap[m] = static_cast<const char*>(vt_) + (((int64_t)b * V + min(m * 16 + l15, V - 1)) * D + s0 * 32 + kq * 8) * vsz;   // rows beyond V: a valid row, results unused
    const float* qp = qt + ((int64_t)b * Q + min(l15, Q - 1)) * D + s0 * 32 + kq * 8;
    lf32x4 acc[MT][GT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < GT; ++g) acc[m][g] = lf32x4{0.f, 0.f, 0.f, 0.f};
    float4 aA[MT][2], qA[2], aB[MT][2], qB[2];
#define CTI_BK_LOAD(a_, q_, s_)                                                                                  \
    {                                                                                                            \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                         \
            a_[m][0] = *reinterpret_cast<const float4*>(ap[m] + (int64_t)(s_) * 32 * vsz);                       \
            if (!VT16) a_[m][1] = *reinterpret_cast<const float4*>(ap[m] + (int64_t)(s_) * 32 * vsz + 16);       \
        }                                                                                                        \
        q_[0] = *reinterpret_cast<const float4*>(qp + (s_) * 32); q_[1] = *reinterpret_cast<const float4*>(qp + (s_) * 32 + 4); \
    }
#define CTI_BK_COMPUTE(a_, q_, s_)                                                                               \
    {                                                                                                            \
        lbf16x8 ah[MT], al[MT];                                                                                  \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                         \
            if (VT16) { ah[m] = __builtin_bit_cast(lbf16x8, a_[m][0]); al[m] = ah[m]; }                          \
            else split8t<TERMS>(a_[m][0], a_[m][1], ah[m], al[m]);                                               \
        }                                                                                                        \
        _Pragma("unroll") for (int g = 0; g < GT; ++g) {                                                         \
            if (g < G) {                                                                                         \
                const float4 h0 = *reinterpret_cast<const float4*>(hl + g * hw + (s_) * 32 + kq * 8);            \
                const float4 h1 = *reinterpret_cast<const float4*>(hl + g * hw + (s_) * 32 + kq * 8 + 4);        \
                lbf16x8 bh, bl;                                                                                  \
                split8t<TERMS>(make_float4(h0.x * q_[0].x, h0.y * q_[0].y, h0.z * q_[0].z, h0.w * q_[0].w),      \
                               make_float4(h1.x * q_[1].x, h1.y * q_[1].y, h1.z * q_[1].z, h1.w * q_[1].w), bh, bl); \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                 \
                    if (TERMS == 3) {                                                                            \
                        if (!VT16) acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh, acc[m][g], 0, 0, 0); \
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl, acc[m][g], 0, 0, 0);      \
                    }                                                                                            \
                    acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh, acc[m][g], 0, 0, 0);          \
                }                                                                                                \
            }                                                                                                    \
        }                                                                                                        \
    }
    if (my > 0) CTI_BK_LOAD(aA, qA, 0)
    for (int s = 0; s < my; s += 2) {
        if (s + 1 < my) CTI_BK_LOAD(aB, qB, s + 1)
        CTI_BK_COMPUTE(aA, qA, s)
        if (s + 1 < my) {
            if (s + 2 < my) CTI_BK_LOAD(aA, qA, s + 2)
            CTI_BK_COMPUTE(aB, qB, s + 1)
        }
    }
#undef CTI_BK_LOAD
#undef CTI_BK_COMPUTE
    // the eight partial outputs, half of the glimpses at a time: part[wave][tile = m * 4 + (g & 3)][reg][lane] (64 lanes x 4 B per register: conflict-free)
    const float hs = h_scale ? h_scale[0] : 1.f;
    const int n = V * Q;
    for (int ph = 0; ph < 2; ++ph) {
        __syncthreads();                                             // (first pass: every wave is done with its h slice; second: the sums of the first are read)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const lf32x4 a0 = acc[m][gg], a1 = acc[m][4 + gg];
                    ksm[((wid * 12 + m * 4 + gg) * 4 + e) * 64 + lane] = ph ? a1[e] : a0[e];
                }
        __syncthreads();
        const int g_lo = ph * 4, g_n = min(4, G - g_lo);
        for (int idx = t; idx < g_n * n; idx += 512) {
            const int gg = idx / n, r = idx - gg * n, v = r / Q, q = r - v * Q;
            const int m = v >> 4, vr = v & 15;
            const float* pp = ksm + ((m * 4 + gg) * 4 + (vr & 3)) * 64 + (vr >> 2) * 16 + q;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += pp[w * 12 * 4 * 64];
            const int g = g_lo + gg;
            logits[((int64_t)b * G + g) * n + r] = sum * hs + (h_bias ? h_bias[g] : 0.f);
        }
    }
}

// =====================================================================================================
// Tri pool on the MFMA (fp32-grade 3-product bf16 mode; A = 3 or 6, Q <= 16, V <= 64, D % 32 == 0).  Per sample:
//   U[v, d] = sum_{(q,a)} w[v,(q,a)] * P[(q,a), d],  P = qt[q,d] * at[a,d];     out[d] = sum_v vt[v,d] * U[v,d]
// The V*Q*A FMAs per channel that bound the VALU forms (27 % of the HBM roofline) become Q*A/16 MFMA steps per 32 x 32 tile.
// M = v (the compacted attention rows, fragments from LDS, loaded ONCE per wave and reused for all its channel tiles),
// N = d: lane = channel, so qt / at / vt are read as coalesced 128-B rows and P is formed per lane; with rows = v in the
// accumulator the Hadamard with vt and the sum over v are in-lane (+ one exchange between the two k-halves).
// =====================================================================================================
template <int A_, int KS, int TERMS, int VT16>
__global__ __launch_bounds__(512) void tri_pool_mfma_kernel(const float* __restrict__ vt, const float* __restrict__ qt,
                                                            const float* __restrict__ at, const float* __restrict__ w,
                                                            int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa,
                                                            float* __restrict__ out, int V, int Q, int D, int tiles_per_wave, int v_rep, PoolShift sh) {
    constexpr int vsz = VT16 ? 2 : 4;                               // (round 5) bytes per element of vt: fp32 or bf16 rows
    // Round 4: the sample's compacted attention slice is split into bf16 hi / lo ONCE per workgroup, into LDS, and the MFMA fragments are read from there at
    // every use (20 ds_read_b128 per tile) instead of living in 80 registers per wave: eight waves fit the register file (two per SIMD), ONE workgroup per sample
    // covers its 32 channel tiles, and the set-up -- what this kernel's time is made of -- is paid once per sample.
    constexpr int QAP = KS * 16, MT = 2, WPH = QAP + 8;            // row pitch in bf16: (QAP + 8) * 2 B = an odd number of 16-B units -> conflict-free ds_read_b128
    extern __shared__ __attribute__((aligned(16))) unsigned short wsm[];     // Wh[64][WPH], Wl[64][WPH]
    unsigned short* const WH = wsm;
    unsigned short* const WL = wsm + 64 * WPH;
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const int QA = Q * A_;
    const float* wb = w + (int64_t)b * w_sb;
    const int nthr = blockDim.x, nwave = blockDim.x >> 6;
    const bool two = V > 32;
    const int tile0 = (blockIdx.x * nwave + wid) * tiles_per_wave;
    const int ntile = min(tiles_per_wave, D / 32 - tile0);
    // The per-tile operands (Q + A + up to 20 vt values per lane, all coalesced 128-B rows) are loaded one tile AHEAD into the other
    // register set: a wave's loads, P formation, MFMA chain and Hadamard would otherwise run strictly one after the other.
    float qA[16], aA[A_], vA[16], wA[16], qB[16], aB[A_], vB[16], wB[16];
#define CTI_TPM_LOAD(qr_, ar_, v0_, v1_, tile_)                                                                  \
    {                                                                                                            \
        const int d_ = (tile_) * 32 + l31;                                                                       \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) qr_[q] = q < Q ? qt[((int64_t)b * Q + q) * D + d_] : 0.f; \
        _Pragma("unroll") for (int a = 0; a < A_; ++a) ar_[a] = at[((int64_t)b * A_ + a) * D + d_];              \
        if (sh.relu || sh.qadd || sh.aadd) {                                                                     \
            const float dq_ = sh.qadd ? sh.qadd[(int64_t)b * D + d_] : 0.f, da_ = sh.aadd ? sh.aadd[(int64_t)b * D + d_] : 0.f; \
            _Pragma("unroll") for (int q = 0; q < 16; ++q) if (q < Q) qr_[q] = shift1(qr_[q], dq_, sh.relu);     \
            _Pragma("unroll") for (int a = 0; a < A_; ++a) ar_[a] = shift1(ar_[a], da_, sh.relu);                \
        }                                                                                                        \
        const char* vb_ = reinterpret_cast<const char*>(vt) + ((int64_t)(b / v_rep) * V * D + d_) * vsz;      /* v_rep > 1: one vt block per image, v_rep batch rows share it */ \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {               /* VT16 is a template argument: no per-load choice of the row format */ \
            const int v = (e & 3) + 8 * (e >> 2) + 4 * kg;                                                       \
            if (VT16) {          /* bf16 rows: a lane PAIR loads the dword that holds both its channels and each lane keeps its half (2-B loads measured 1.7x slower) */ \
                const char* vp_ = vb_ - (l31 & 1) * 2;                                                           \
                const unsigned x0_ = v < V ? *reinterpret_cast<const unsigned*>(vp_ + (int64_t)v * D * 2) : 0u;  \
                const unsigned x1_ = (two && v + 32 < V) ? *reinterpret_cast<const unsigned*>(vp_ + (int64_t)(v + 32) * D * 2) : 0u; \
                v0_[e] = __builtin_bit_cast(float, (x0_ << ((l31 & 1) ? 0 : 16)) & 0xffff0000u);                 \
                v1_[e] = __builtin_bit_cast(float, (x1_ << ((l31 & 1) ? 0 : 16)) & 0xffff0000u);                 \
            } else {                                                                                             \
                v0_[e] = v < V ? *reinterpret_cast<const float*>(vb_ + (int64_t)v * D * 4) : 0.f;                \
                v1_[e] = (two && v + 32 < V) ? *reinterpret_cast<const float*>(vb_ + (int64_t)(v + 32) * D * 4) : 0.f; \
            }                                                                                                    \
        }                                                                                                        \
    }
#ifndef CTI_TPM_ABL
#define CTI_TPM_ABL 0        // timing-only ablations (wrong results): 1 = set-up only (no tile loop), 2 = no gather of the attention slice (constants into LDS), 4 = no vt loads
#endif
    if (ntile > 0) CTI_TPM_LOAD(qA, aA, vA, wA, tile0)              // (in flight while the attention slice is compacted and split below)
    for (int it = t; it < 64 * (QAP / 8); it += nthr) {             // (row v, eight consecutive (q, a) columns)
        const int v = it / (QAP / 8), c0 = (it - v * (QAP / 8)) * 8;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int qa = c0 + e, q = qa / A_, a = qa - q * A_;
            x[e] = (CTI_TPM_ABL & 2) ? 0.01f * qa : ((v < V && qa < QA) ? wb[v * w_sv + q * w_sq + a * w_sa] : 0.f);
        }
        lbf16x8 hi, lo;
        split8t<TERMS>(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), hi, lo);       // TERMS = 1 (plain-bf16 mode): hi only, one product per pair
        *reinterpret_cast<lbf16x8*>(WH + v * WPH + c0) = hi;
        if (TERMS == 3) *reinterpret_cast<lbf16x8*>(WL + v * WPH + c0) = lo;
    }
    __syncthreads();
#define CTI_TPM_COMPUTE(qr_, ar_, v0_, v1_, tile_)                                                               \
    {                                                                                                            \
        lf32x16 u0, u1;                                                                                          \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) { u0[e] = 0.f; u1[e] = 0.f; }                             \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                      \
            float p0[8], p1[8];                                                                                  \
            _Pragma("unroll") for (int uu = 0; uu < 8; ++uu) {                                                   \
                const int k0 = ks * 16 + uu, k1 = ks * 16 + 8 + uu;                                              \
                p0[uu] = (k0 / A_ < 16) ? qr_[(k0 / A_) & 15] * ar_[k0 % A_] : 0.f;                              \
                p1[uu] = (k1 / A_ < 16) ? qr_[(k1 / A_) & 15] * ar_[k1 % A_] : 0.f;                              \
            }                                                                                                    \
            const float4 pa = kg ? make_float4(p1[0], p1[1], p1[2], p1[3]) : make_float4(p0[0], p0[1], p0[2], p0[3]); \
            const float4 pb = kg ? make_float4(p1[4], p1[5], p1[6], p1[7]) : make_float4(p0[4], p0[5], p0[6], p0[7]); \
            lbf16x8 ph, pl;                                                                                      \
            split8t<TERMS>(pa, pb, ph, pl);                                                                      \
            const int wo_ = l31 * WPH + ks * 16 + kg * 8;                                                        \
            const lbf16x8 wh0 = *reinterpret_cast<const lbf16x8*>(WH + wo_);                                     \
            lbf16x8 wl0 = wh0;                                                                                   \
            if (TERMS == 3) wl0 = *reinterpret_cast<const lbf16x8*>(WL + wo_);                                   \
            lbf16x8 wh1 = wh0, wl1 = wl0;                                                                        \
            if (two) { wh1 = *reinterpret_cast<const lbf16x8*>(WH + wo_ + 32 * WPH); if (TERMS == 3) wl1 = *reinterpret_cast<const lbf16x8*>(WL + wo_ + 32 * WPH); } \
            if (TERMS == 3) {                                                                                    \
            u0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl0, ph, u0, 0, 0, 0);                            \
            if (two) u1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl1, ph, u1, 0, 0, 0);                   \
            u0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh0, pl, u0, 0, 0, 0);                            \
            if (two) u1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh1, pl, u1, 0, 0, 0);                   \
            }                                                                                                    \
            u0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh0, ph, u0, 0, 0, 0);                            \
            if (two) u1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh1, ph, u1, 0, 0, 0);                   \
            __builtin_amdgcn_sched_barrier(0);     /* (the scheduler would hoist all 20 fragment reads of a tile: 80 registers) */ \
        }                                                                                                        \
        float acc = 0.f;                                                                                         \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) { acc = fmaf(u0[e], v0_[e], acc); acc = fmaf(u1[e], v1_[e], acc); } \
        acc += __shfl_xor(acc, 32, 64);                                                                          \
        if (kg == 0) out[(int64_t)b * D + (tile_) * 32 + l31] = acc;                                             \
    }
    for (int ti = 0; ti < ((CTI_TPM_ABL & 1) ? min(ntile, 1) : ntile); ti += 2) {
        if (ti + 1 < ntile) CTI_TPM_LOAD(qB, aB, vB, wB, tile0 + ti + 1)
        CTI_TPM_COMPUTE(qA, aA, vA, wA, tile0 + ti)
        if (ti + 1 < ntile) {
            if (ti + 2 < ntile) CTI_TPM_LOAD(qA, aA, vA, wA, tile0 + ti + 2)
            CTI_TPM_COMPUTE(qB, aB, vB, wB, tile0 + ti + 1)
        }
    }
#undef CTI_TPM_LOAD
#undef CTI_TPM_COMPUTE
}

// =====================================================================================================
// Attention-gradient of the sum-pools on the MFMA (fp32-grade mode):  dw[b,v,(q,a)] = sum_d (dout[b,d] vt[b,v,d]) (qt[b,q,d] at[b,a,d])
// (bi pool: at == NULL, columns are q).  Per sample a (V x D) . (Q*A x D)^T contraction with the channel axis d contiguous in every
// operand: fragments straight from global memory, both operands formed on the fly (two element-wise products), no LDS.  The VALU form
// spends its time in 32 wave reductions per (b, v) workgroup (350 us at B = 256); this one ~1/10 of that.
// =====================================================================================================
__global__ __launch_bounds__(256) void pool_dw_mfma_kernel(const float* __restrict__ dout, const float* __restrict__ vt, const float* __restrict__ qt,
                                                           const float* __restrict__ at, float* __restrict__ dw, int V, int Q, int A, int D,
                                                           int MT, int NT, int dper) {
    __shared__ float red[3][16][64];
    const int b = blockIdx.x, ks = blockIdx.y;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 31, kg = lane >> 5;
    const int N = Q * A;
    const int ntiles = MT * NT;
    // with one or two tiles per sample (the bi pools: V <= 64, Q <= 16) the spare waves take further slices of this workgroup's channel range
    // for the SAME tiles and hand their accumulators over through LDS (fixed order), instead of idling
    const int nsub = ntiles == 1 ? 4 : (ntiles == 2 ? 2 : 1);
    const int sub = nsub > 1 ? wid / ntiles : 0;
    const int wg_lo = ks * dper, wg_hi = min(D, wg_lo + dper);
    const int sper = nsub > 1 ? (((wg_hi - wg_lo) / 16 + nsub - 1) / nsub) * 16 : (wg_hi - wg_lo);
    const int d_lo = wg_lo + sub * sper, d_hi = min(wg_hi, d_lo + sper);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tile = nsub > 1 ? wid % ntiles : wid; tile < ntiles; tile += 4) {
        const int mt = tile % MT, nt = tile / MT;
        const int v = mt * 32 + r, c = nt * 32 + r;
        const bool vok = v < V, cok = c < N;
        const int q = cok ? c / A : 0, a = cok ? c - q * A : 0;
        const float* vp = vt + ((int64_t)b * V + (vok ? v : 0)) * D + kg * 8;
        const float* gp = dout + (int64_t)b * D + kg * 8;
        const float* qp = qt + ((int64_t)b * Q + q) * D + kg * 8;
        const float* ap = at ? at + ((int64_t)b * A + a) * D + kg * 8 : nullptr;
        lf32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int d0 = d_lo; d0 < d_hi; d0 += 16) {
            float4 x0 = z4, x1 = z4, p0 = z4, p1 = z4;
            if (vok) {
                const float4 a0 = *reinterpret_cast<const float4*>(vp + d0), a1 = *reinterpret_cast<const float4*>(vp + d0 + 4);
                const float4 g0 = *reinterpret_cast<const float4*>(gp + d0), g1 = *reinterpret_cast<const float4*>(gp + d0 + 4);
                x0 = make_float4(a0.x * g0.x, a0.y * g0.y, a0.z * g0.z, a0.w * g0.w);
                x1 = make_float4(a1.x * g1.x, a1.y * g1.y, a1.z * g1.z, a1.w * g1.w);
            }
            if (cok) {
                p0 = *reinterpret_cast<const float4*>(qp + d0); p1 = *reinterpret_cast<const float4*>(qp + d0 + 4);
                if (ap) {
                    const float4 t0 = *reinterpret_cast<const float4*>(ap + d0), t1 = *reinterpret_cast<const float4*>(ap + d0 + 4);
                    p0 = make_float4(p0.x * t0.x, p0.y * t0.y, p0.z * t0.z, p0.w * t0.w);
                    p1 = make_float4(p1.x * t1.x, p1.y * t1.y, p1.z * t1.z, p1.w * t1.w);
                }
            }
            lbf16x8 xh, xl, ph, pl;
            split8(x0, x1, xh, xl);
            split8(p0, p1, ph, pl);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, ph, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, pl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ph, acc, 0, 0, 0);
        }
        if (nsub > 1) {                                              // every wave runs this loop body exactly once in that case
            if (sub > 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) red[wid - ntiles][e][lane] = acc[e];
            }
            __syncthreads();
            if (sub == 0) {
                for (int s2 = 1; s2 < nsub; ++s2)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] += red[s2 * ntiles + tile - ntiles][e][lane];
            }
        }
        if (cok && sub == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int vv = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
                if (vv < V) atomicAdd(dw + ((int64_t)b * V + vv) * N + c, acc[e]);
            }
        }
    }
}

// =====================================================================================================
// Bilinear-logits backward on the MFMA (fp32-grade mode; G <= 8, V <= 64, Q <= 16, D % 32 == 0).  With U[g] = dlogits[b,g] (V x Q):
//   dvt[v,d] = hs sum_g h[g,d] (U[g] qt)[v,d]       dh_b[g,d] = hs sum_v vt[v,d] (U[g] qt)[v,d]       (kernel 1: contraction over q)
//   dqt[q,d] = hs sum_g h[g,d] (U[g]^T vt)[q,d]                                                        (kernel 2: contraction over v)
// Lane = channel d (N axis): qt / vt / h are read as coalesced rows and turned into B fragments per lane; U (16 KiB per sample) is staged
// once per workgroup in LDS, zero-padded to the fragment shapes, and provides the A fragments; the per-g products with h[g,d] and the
// sums over v are in-lane.  The VALU kernel keeps V + Q accumulators per channel in LDS and walks G*V*Q terms per channel: 3.4 ms at
// B = 256, G = 8, D = 3072.
// =====================================================================================================
constexpr int BLG = 8;       // G bound of these kernels

__global__ __launch_bounds__(256) void bi_logits_bwd_vh_kernel(const float* __restrict__ dl, const float* __restrict__ vt, const float* __restrict__ qt,
                                                               const float* __restrict__ h, const float* __restrict__ h_scale,
                                                               float* __restrict__ dvt, float* __restrict__ dhpart, int G, int V, int Q, int D,
                                                               int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // Us[g][64 v][16 q (+4 pad)]
    constexpr int UP = 20;
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const float* dlb = dl + (int64_t)b * G * V * Q;
    for (int i = t; i < G * 64 * 16; i += 256) {
        const int q = i & 15, v = (i >> 4) & 63, g = i >> 10;
        sm[(g * 64 + v) * UP + q] = (v < V && q < Q) ? dlb[((int64_t)g * V + v) * Q + q] : 0.f;
    }
    __syncthreads();
    const float hs = h_scale ? h_scale[0] : 1.f;
    const bool two = V > 32;
    const int tile0 = (blockIdx.x * 4 + wid) * tiles_per_wave;
    for (int ti = 0; ti < tiles_per_wave; ++ti) {
        const int d0 = (tile0 + ti) * 32;
        if (d0 >= D) break;
        const int d = d0 + l31;
        // B fragment: qt[q = kg*8 .. +8, d]
        float qv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) qv[u] = (kg * 8 + u < Q) ? qt[((int64_t)b * Q + kg * 8 + u) * D + d] : 0.f;
        lbf16x8 qh, ql;
        split8(make_float4(qv[0], qv[1], qv[2], qv[3]), make_float4(qv[4], qv[5], qv[6], qv[7]), qh, ql);
        float v0[16], v1[16];
        const float* vb = vt + (int64_t)b * V * D + d;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int v = (e & 3) + 8 * (e >> 2) + 4 * kg;
            v0[e] = v < V ? vb[(int64_t)v * D] : 0.f;
            v1[e] = (two && v + 32 < V) ? vb[(int64_t)(v + 32) * D] : 0.f;
        }
        lf32x16 o0, o1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
        for (int g = 0; g < G; ++g) {
            const float hg = h[(int64_t)g * D + d];
            const float* u0 = sm + (g * 64 + l31) * UP + kg * 8;
            lbf16x8 ah, al;
            split8(*reinterpret_cast<const float4*>(u0), *reinterpret_cast<const float4*>(u0 + 4), ah, al);
            lf32x16 w0, w1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { w0[e] = 0.f; w1[e] = 0.f; }
            // four products here (lo*lo too): dh_b feeds the weight-norm gain gradient <G, V> / g, a cancelling sum that amplifies the 2^-16 of
            // the three-product form past the 1e-4 bar on small layers; the MFMA time of this kernel is negligible either way
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ql, w0, 0, 0, 0);
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh, w0, 0, 0, 0);
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql, w0, 0, 0, 0);
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh, w0, 0, 0, 0);
            if (two) {
                const float* u1 = u0 + 32 * UP;
                lbf16x8 bh, bl;
                split8(*reinterpret_cast<const float4*>(u1), *reinterpret_cast<const float4*>(u1 + 4), bh, bl);
                w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ql, w1, 0, 0, 0);
                w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, qh, w1, 0, 0, 0);
                w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ql, w1, 0, 0, 0);
                w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, qh, w1, 0, 0, 0);
            }
            float sacc = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                o0[e] = fmaf(hg, w0[e], o0[e]); o1[e] = fmaf(hg, w1[e], o1[e]);
                sacc = fmaf(w0[e], v0[e], sacc); sacc = fmaf(w1[e], v1[e], sacc);
            }
            sacc += __shfl_xor(sacc, 32, 64);
            if (kg == 0) dhpart[((int64_t)b * G + g) * D + d] = hs * sacc;
        }
        float* ob = dvt + (int64_t)b * V * D + d;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int v = (e & 3) + 8 * (e >> 2) + 4 * kg;
            if (v < V) ob[(int64_t)v * D] = hs * o0[e];
            if (two && v + 32 < V) ob[(int64_t)(v + 32) * D] = hs * o1[e];
        }
    }
}

__global__ __launch_bounds__(256) void bi_logits_bwd_q_kernel(const float* __restrict__ dl, const float* __restrict__ vt, const float* __restrict__ h,
                                                              const float* __restrict__ h_scale, float* __restrict__ dqt, int G, int V, int Q, int D,
                                                              int tiles_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // Ut[g][32 q][64 v (+4 pad)]
    constexpr int TPv = 68;
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const float* dlb = dl + (int64_t)b * G * V * Q;
    for (int i = t; i < G * 32 * 64; i += 256) {
        const int v = i & 63, q = (i >> 6) & 31, g = i >> 11;
        sm[(g * 32 + q) * TPv + v] = (v < V && q < Q) ? dlb[((int64_t)g * V + v) * Q + q] : 0.f;
    }
    __syncthreads();
    const float hs = h_scale ? h_scale[0] : 1.f;
    const int KS = (V + 15) / 16;                                      // 16-deep steps over v
    const int tile0 = (blockIdx.x * 4 + wid) * tiles_per_wave;
    for (int ti = 0; ti < tiles_per_wave; ++ti) {
        const int d0 = (tile0 + ti) * 32;
        if (d0 >= D) break;
        const int d = d0 + l31;
        const float* vb = vt + (int64_t)b * V * D + d;
        lbf16x8 vh[4], vl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int v = ks * 16 + kg * 8 + u; x[u] = (ks < KS && v < V) ? vb[(int64_t)v * D] : 0.f; }
            split8(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), vh[ks], vl[ks]);
        }
        lf32x16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
        for (int g = 0; g < G; ++g) {
            const float hg = h[(int64_t)g * D + d];
            lf32x16 w;
#pragma unroll
            for (int e = 0; e < 16; ++e) w[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < KS) {
                    const float* u0 = sm + (g * 32 + l31) * TPv + ks * 16 + kg * 8;
                    lbf16x8 ah, al;
                    split8(*reinterpret_cast<const float4*>(u0), *reinterpret_cast<const float4*>(u0 + 4), ah, al);
                    w = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, vl[ks], w, 0, 0, 0);     // four products, as in the dvt kernel
                    w = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, vh[ks], w, 0, 0, 0);
                    w = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, vl[ks], w, 0, 0, 0);
                    w = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, vh[ks], w, 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = fmaf(hg, w[e], o[e]);
        }
        float* ob = dqt + (int64_t)b * Q * D + d;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int q = (e & 3) + 8 * (e >> 2) + 4 * kg;
            if (q < Q) ob[(int64_t)q * D] = hs * o[e];
        }
    }
}

}  // namespace
}  // namespace cti

using namespace cti;

// chunking of the Tri softmax: chunks of <= 32768 positions (x G floats), at least one per sample
static void tri_chunks(int V, int64_t QA, int64_t* chunk_n, int* nchunk) {
    const int64_t N = (int64_t)V * QA;
    const int64_t c = tuning_tri_chunk() > 0 ? tuning_tri_chunk() : 32768;
    *nchunk = (int)((N + c - 1) / c);
    *chunk_n = c;
}

extern "C" size_t cti_softmax_tri_workspace_bytes(int B, int V, int64_t QA, int G) {
    if (B <= 0 || V <= 0 || QA <= 0 || G <= 0) return 0;
    int64_t cn; int nc; tri_chunks(V, QA, &cn, &nc);
    return sizeof(float) * 2 * ((size_t)B * nc * G + (size_t)B * G);
}

extern "C" int cti_masked_softmax_tri_fwd(float* logits, const uint8_t* mask, float* p, int B, int V, int64_t QA, int G,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(logits); CTI_REQUIRE_PTR(mask); CTI_REQUIRE_PTR(p); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(B > 0 && V > 0 && QA > 0 && G > 0 && B <= 65535, CTI_E_SHAPE, "cti_masked_softmax_tri_fwd: B=%d V=%d QA=%lld G=%d",
                B, V, (long long)QA, G);
    CTI_REQUIRE(workspace_bytes >= cti_softmax_tri_workspace_bytes(B, V, QA, G), CTI_E_WORKSPACE,
                "cti_masked_softmax_tri_fwd: workspace %zu < %zu", workspace_bytes, cti_softmax_tri_workspace_bytes(B, V, QA, G));
    int64_t cn; int nc; tri_chunks(V, QA, &cn, &nc);
    float* part = static_cast<float*>(workspace);
    float* stats = part + (size_t)B * nc * G * 2;
    hipStream_t st = as_stream(stream);
    dim3 grid(nc, B);
    const bool al16 = ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(p)) & 15) == 0;
    const bool fast2 = G == 2 && al16 && (((int64_t)V * QA) & 1) == 0 && (cn & 1) == 0;       // even position counts: every sample starts 16-B aligned
    if (fast2) hipLaunchKernelGGL(tri_partial_g2_kernel, grid, dim3(SM_THREADS), 0, st, logits, mask, part, V, QA, cn, nc);
    else
    switch (G) {
#define CTI_CASE(g) case g: hipLaunchKernelGGL(tri_partial_kernel<g>, grid, dim3(SM_THREADS), 0, st, logits, mask, part, V, QA, G, cn, nc); break;
        CTI_CASE(1) CTI_CASE(2) CTI_CASE(3) CTI_CASE(4) CTI_CASE(8)
#undef CTI_CASE
        default: hipLaunchKernelGGL(tri_partial_kernel<0>, grid, dim3(SM_THREADS), 0, st, logits, mask, part, V, QA, G, cn, nc);
    }
    int rc = launch_status("cti_masked_softmax_tri_fwd/partial"); if (rc) return rc;
    hipLaunchKernelGGL(tri_combine_kernel, dim3(B, G), dim3(64), 0, st, part, stats, G, nc);
    rc = launch_status("cti_masked_softmax_tri_fwd/combine"); if (rc) return rc;
    const int64_t NG = (int64_t)V * QA * G;
    const int64_t nblk = (NG + 255) / 256;
    const unsigned gx = (unsigned)(nblk < 2048 ? nblk : 2048);
    if (al16 && (G == 1 || G == 2 || G == 4) && (NG & 3) == 0) {
        const int64_t nb4 = (NG / 4 + 255) / 256;
        hipLaunchKernelGGL(tri_normalise_v4_kernel, dim3((unsigned)(nb4 < 4096 ? nb4 : 4096), B), dim3(256), 0, st, logits, stats, p, NG / 4, G);
    } else {
        hipLaunchKernelGGL(tri_normalise_kernel, dim3(gx, B), dim3(256), 0, st, logits, stats, p, NG, G);
    }
    return launch_status("cti_masked_softmax_tri_fwd/normalise");
}

extern "C" int cti_masked_softmax_tri_from_partials_fwd(float* logits, const uint8_t* mask, const float* partials, size_t partials_bytes, float* p,
                                                        int B, int V, int64_t QA, int G, void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(logits); CTI_REQUIRE_PTR(mask); CTI_REQUIRE_PTR(partials); CTI_REQUIRE_PTR(p); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(B > 0 && V > 0 && QA > 0 && B <= 65535, CTI_E_SHAPE, "cti_masked_softmax_tri_from_partials_fwd: B=%d V=%d QA=%lld", B, V, (long long)QA);
    CTI_REQUIRE(G == 2, CTI_E_UNSUPPORTED, "cti_masked_softmax_tri_from_partials_fwd: G=%d (the fused partial pass exists for glimpse 2)", G);
    const size_t per = sizeof(float) * (size_t)B * G * 2;
    CTI_REQUIRE(partials_bytes >= per && partials_bytes % per == 0, CTI_E_SHAPE, "cti_masked_softmax_tri_from_partials_fwd: %zu bytes of partials are not a whole number of (B, G) pairs", partials_bytes);
    CTI_REQUIRE(workspace_bytes >= per, CTI_E_WORKSPACE, "cti_masked_softmax_tri_from_partials_fwd: workspace %zu < %zu", workspace_bytes, per);
    CTI_REQUIRE(((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(p)) & 15) == 0 && (((int64_t)V * QA) & 1) == 0, CTI_E_ALIGN,
                "cti_masked_softmax_tri_from_partials_fwd: logits / p must be 16-B aligned and V*QA even");
    const int nchunk = (int)(partials_bytes / per);
    float* stats = static_cast<float*>(workspace);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(tri_combine_kernel, dim3(B, G), dim3(64), 0, st, partials, stats, G, nchunk);
    int rc = launch_status("cti_masked_softmax_tri_from_partials_fwd/combine"); if (rc) return rc;
    const int64_t N = (int64_t)V * QA;
    int64_t cn = tuning_tri_chunk() > 0 ? (tuning_tri_chunk() + 1) & ~(int64_t)1 : 2048;      // positions per workgroup (even).  Measured at C2, B = 64: 2048 -> 328 us, 16384 -> 376, 65536 -> 403 (a plain copy of the logits: 308)
    const int64_t nc = (N + cn - 1) / cn;
    hipLaunchKernelGGL(tri_normalise_mask_g2_kernel, dim3((unsigned)nc, B), dim3(SM_THREADS), 0, st, logits, mask, stats, p, V, QA, cn);
    return launch_status("cti_masked_softmax_tri_from_partials_fwd/normalise");
}

extern "C" int cti_masked_softmax_bi_fwd(float* logits, const uint8_t* mask, float* p, int B, int G, int V, int Q, void* stream) {
    CTI_REQUIRE_PTR(logits); CTI_REQUIRE_PTR(p);
    CTI_REQUIRE(B > 0 && G > 0 && V > 0 && Q > 0, CTI_E_SHAPE, "cti_masked_softmax_bi_fwd: B=%d G=%d V=%d Q=%d", B, G, V, Q);
    const int rows = B * G;
    hipLaunchKernelGGL(bi_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), logits, mask, p, rows, G, V, Q);
    return launch_status("cti_masked_softmax_bi_fwd");
}

// sh set (the shifted form): only the table / streaming kernels take it -- any other shape returns CTI_E_UNSUPPORTED before anything is launched
static int tri_pool_impl(const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                         int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, PoolShift sh, void* stream) {
    const bool shifted = sh.relu || sh.qadd || sh.aadd;
    CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(at); CTI_REQUIRE_PTR(w); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && A > 0 && D > 0 && B <= 65535, CTI_E_SHAPE, "cti_tri_pool_fwd: B=%d V=%d Q=%d A=%d D=%d", B, V, Q, A, D);
    if (shifted && ((sh.qadd && !aligned16(sh.qadd)) || (sh.aadd && !aligned16(sh.aadd)))) return CTI_E_UNSUPPORTED;
#ifndef CTI_TRI_TABLE
#define CTI_TRI_TABLE 1
#endif
#ifndef CTI_TRI_TABLE_NG
#define CTI_TRI_TABLE_NG 1
#endif
    if (CTI_TRI_TABLE && A <= 8 && D % 4 == 0 && aligned16(vt) && aligned16(qt) && aligned16(at) && aligned16(out)) {
        const int QA = Q * A;
        constexpr int NGT = CTI_TRI_TABLE_NG;
        const dim3 grid((D / 2 + 127) / 128, B);
        size_t lds_t = sizeof(float) * (size_t)V * ((QA + 3) & ~3);
        if (lds_t < sizeof(float2) * 128 * (NGT - 1)) lds_t = sizeof(float2) * 128 * (NGT - 1);
#define CTI_TT(QAv, Av) hipLaunchKernelGGL((tri_pool_table_kernel<QAv, NGT, Av>), grid, dim3(128 * NGT), lds_t, as_stream(stream), vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, V, Q, A, D, sh)
        if (lds_t <= 64 * 1024 && A == 3) {
            // measured at B = 256, D = 1024 (rocprofv3): QA = 42: 27.1 us (stream form 31.6); QA = 72: 53 us (stream form 46.5, kept there)
            if (QA == 42) { CTI_TT(42, 3); return launch_status("cti_tri_pool_fwd"); }
            if (QA == 36) { CTI_TT(36, 3); return launch_status("cti_tri_pool_fwd"); }
        }
#undef CTI_TT
    }
    if (A <= 8 && Q <= 16 && D % 4 == 0 && aligned16(vt) && aligned16(qt) && aligned16(at) && aligned16(out)) {
        const int AP = A <= 4 ? 4 : 8;
#ifndef CTI_POOL_NG
#define CTI_POOL_NG 2
#endif
        constexpr int NG = CTI_POOL_NG;
        size_t lds_p = sizeof(float) * (size_t)V * Q * AP;
        if (lds_p < sizeof(float4) * 128 * (NG - 1)) lds_p = sizeof(float4) * 128 * (NG - 1);
        if (lds_p <= 64 * 1024) {
            const dim3 grid((D / 4 + 127) / 128, B);
#define CTI_TP(APv, QXv) hipLaunchKernelGGL((pool_stream_kernel<true, APv, NG, QXv>), grid, dim3(128 * NG), lds_p, as_stream(stream), vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, V, Q, A, D, sh, 0, PoolAdds{})
            if (AP == 4) { if (Q == 14) CTI_TP(4, 14); else if (Q == 12) CTI_TP(4, 12); else CTI_TP(4, 0); }
            else         { if (Q == 14) CTI_TP(8, 14); else if (Q == 12) CTI_TP(8, 12); else CTI_TP(8, 0); }
#undef CTI_TP
            return launch_status("cti_tri_pool_fwd");
        }
    }
    if (shifted) return CTI_E_UNSUPPORTED;
    if (A <= 8) {
        const int AP = A <= 4 ? 4 : 8;
        const size_t lds_s = sizeof(float) * ((size_t)V * Q * AP + 256 * (size_t)Q);
        if (lds_s <= 64 * 1024) {
            if (AP == 4) hipLaunchKernelGGL(tri_pool_small_kernel<4>, dim3((D + 255) / 256, B), dim3(256), lds_s, as_stream(stream), vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, V, Q, A, D);
            else         hipLaunchKernelGGL(tri_pool_small_kernel<8>, dim3((D + 255) / 256, B), dim3(256), lds_s, as_stream(stream), vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, V, Q, A, D);
            return launch_status("cti_tri_pool_fwd");
        }
    }
    size_t lds = sizeof(float) * 256 * (size_t)(Q + A);
    int stage_a = 1;
    if (lds > 64 * 1024) { stage_a = 0; lds = sizeof(float) * 256 * (size_t)Q; }
    CTI_REQUIRE(lds <= 64 * 1024, CTI_E_SHAPE, "cti_tri_pool_fwd: Q=%d too large for the LDS staging", Q);
    hipLaunchKernelGGL(tri_pool_kernel, dim3((D + 255) / 256, B), dim3(256), lds, as_stream(stream), vt, qt, at, w, w_sb, w_sv,
                       w_sq, w_sa, out, V, Q, A, D, stage_a);
    return launch_status("cti_tri_pool_fwd");
}

extern "C" int cti_tri_pool_fwd(const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                                int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, void* stream) {
    return tri_pool_impl(vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, B, V, Q, A, D, PoolShift{nullptr, nullptr, 0}, stream);
}

static int bi_pool_impl(const float* vt, const float* qt, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                        float* out, int B, int V, int Q, int D, int k, PoolShift sh, void* stream, int vt16 = 0, const PoolAdds* qa = nullptr) {
    const bool shifted = sh.relu || sh.qadd || qa;
    CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(out);
    if (shifted && sh.qadd && !aligned16(sh.qadd)) return CTI_E_UNSUPPORTED;
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && D > 0 && k > 0 && B <= 65535, CTI_E_SHAPE, "cti_bi_pool_fwd: B=%d V=%d Q=%d D=%d k=%d", B, V, Q, D, k);
    CTI_REQUIRE(D >= k, CTI_E_SHAPE, "cti_bi_pool_fwd: D=%d < k=%d", D, k);
    if (k == 1 && Q <= 16 && D % 4 == 0 && (size_t)V * 16 * sizeof(float) <= 64 * 1024 && aligned16(vt) && aligned16(qt) && aligned16(out)) {
#ifndef CTI_POOL_NG_BI
#define CTI_POOL_NG_BI 2
#endif
        constexpr int NGB = CTI_POOL_NG_BI;
        size_t lds_b = sizeof(float) * (size_t)V * 16;
        if (lds_b < sizeof(float4) * 128 * (NGB - 1)) lds_b = sizeof(float4) * 128 * (NGB - 1);
#define CTI_BP(QXv) hipLaunchKernelGGL((pool_stream_kernel<false, 4, NGB, QXv>), dim3((D / 4 + 127) / 128, B), dim3(128 * NGB), lds_b, as_stream(stream), \
                           vt, qt, nullptr, w, w_sb, w_sv, w_sq, (int64_t)0, out, V, Q, 0, D, sh, vt16, qa ? *qa : PoolAdds{})
        if (Q == 14) CTI_BP(14); else if (Q == 12) CTI_BP(12); else CTI_BP(0);
#undef CTI_BP
        return launch_status("cti_bi_pool_fwd");
    }
    if (shifted || vt16) return CTI_E_UNSUPPORTED;                 // (only the k = 1 streaming kernel forms relu(row + add) on load / reads bf16 rows)
    if (k == 3 && Q <= 16 && D % 6 == 0 && (size_t)V * 16 * sizeof(float) <= 64 * 1024 && ((reinterpret_cast<uintptr_t>(vt) | reinterpret_cast<uintptr_t>(qt) | reinterpret_cast<uintptr_t>(out)) & 7) == 0) {
        constexpr int NG3 = 2;
        size_t lds3 = sizeof(float) * (size_t)V * 16;
        if (lds3 < sizeof(float2) * 128 * (NG3 - 1)) lds3 = sizeof(float2) * 128 * (NG3 - 1);
        hipLaunchKernelGGL((bi_pool_k3_kernel<NG3>), dim3((D / 6 + 127) / 128, B), dim3(128 * NG3), lds3, as_stream(stream), vt, qt, w, w_sb, w_sv, w_sq, out, V, Q, D);
        return launch_status("cti_bi_pool_fwd");
    }
    const size_t lds = sizeof(float) * 256 * (size_t)Q * k;
    CTI_REQUIRE(lds <= 64 * 1024, CTI_E_SHAPE, "cti_bi_pool_fwd: Q*k=%d too large for the LDS staging (<= 64)", Q * k);
    const int NO = D / k;                                   // AvgPool1d(k, stride=k) drops a ragged tail, like torch
    hipLaunchKernelGGL(bi_pool_kernel, dim3((NO + 255) / 256, B), dim3(256), lds, as_stream(stream), vt, qt, w, w_sb, w_sv, w_sq,
                       out, V, Q, D, k);
    return launch_status("cti_bi_pool_fwd");
}

extern "C" int cti_bi_pool_fwd(const float* vt, const float* qt, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                               float* out, int B, int V, int Q, int D, int k, void* stream) {
    return bi_pool_impl(vt, qt, w, w_sb, w_sv, w_sq, out, B, V, Q, D, k, PoolShift{nullptr, nullptr, 0}, stream);
}

// out[b,d] = sum_v vt[b,v,d] * sum_q w[b,v,q] * relu(qt[b,q,d] + qadd[b,d])   (k = 1; qadd may be NULL = 0).  CTI_E_UNSUPPORTED when the shape is not the
// streaming kernel's (Q <= 16, D % 4 == 0, 16-B aligned rows): nothing is launched, the caller materialises the operand and takes cti_bi_pool_fwd.
extern "C" int cti_bi_pool_shift_fwd(const float* vt, const float* qt, const float* qadd, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                                     float* out, int B, int V, int Q, int D, void* stream) {
    CTI_REQUIRE_PTR(w);
    return bi_pool_impl(vt, qt, w, w_sb, w_sv, w_sq, out, B, V, Q, D, 1, PoolShift{qadd, nullptr, 1}, stream);
}

// The same with `vt` as bf16 rows (round 5: what the plain-bf16 mode's hoisted v projection writes -- cti_gemm_bf16_rows with c_bf16 = 1).  8-B aligned rows
// (D % 4 == 0); CTI_E_UNSUPPORTED outside the streaming kernel's shapes, as above.
extern "C" int cti_bi_pool_shift_vt16_fwd(const void* vt_bf16, const float* qt, const float* qadd, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                                          float* out, int B, int V, int Q, int D, void* stream) {
    CTI_REQUIRE_PTR(w);
    return bi_pool_impl(static_cast<const float*>(vt_bf16), qt, w, w_sb, w_sv, w_sq, out, B, V, Q, D, 1, PoolShift{qadd, nullptr, 1}, stream, 1);
}

// The shifted bi pool whose shift is a SUM of n_qadd addends, qadds[i] a (B, qadd_ld[i]) fp32 matrix read at columns [0, D) -- row stride 0 = one row for the whole
// batch --: out[b,d] = sum_vq vt[b,v,d] w[b,v,q] relu(qt[b,q,d] + sum_i qadds[i][b * ld_i + d]).  qadds / qadd_ld are HOST arrays (consumed before the call returns);
// every addend 16-B aligned with ld % 4 == 0.  vt_bf16 != 0: vt as bf16 rows.  CTI_E_UNSUPPORTED (nothing launched) outside the streaming kernel's shapes or for more than 32 addends.
extern "C" int cti_bi_pool_shift_multi_fwd(const void* vt, int vt_bf16, const float* qt, const float* const* qadds, const int64_t* qadd_ld, int n_qadd, const float* w,
                                           int64_t w_sb, int64_t w_sv, int64_t w_sq, float* out, int64_t ldo, int B, int V, int Q, int D, void* stream) {
    CTI_REQUIRE_PTR(w);
    CTI_REQUIRE(ldo >= D && (ldo & 3) == 0 && ldo < (1ll << 31), CTI_E_SHAPE, "cti_bi_pool_shift_multi_fwd: ldo=%lld (>= D, a multiple of 4)", (long long)ldo);
    CTI_REQUIRE(n_qadd >= 0 && (n_qadd == 0 || (qadds && qadd_ld)), CTI_E_NULL, "cti_bi_pool_shift_multi_fwd: n_qadd=%d without the arrays", n_qadd);
    if (n_qadd > POOL_MAX_ADDS) return CTI_E_UNSUPPORTED;
    PoolAdds qa{};
    qa.n = n_qadd;
    for (int i = 0; i < n_qadd; ++i) {
        CTI_REQUIRE(qadds[i] != nullptr && qadd_ld[i] >= 0 && qadd_ld[i] < (1ll << 31), CTI_E_SHAPE, "cti_bi_pool_shift_multi_fwd: addend %d", i);
        if (!aligned16(qadds[i]) || (qadd_ld[i] & 3)) return CTI_E_UNSUPPORTED;
        qa.p[i] = qadds[i]; qa.ld[i] = (int)qadd_ld[i];
    }
    qa.ldo = (int)ldo;
    return bi_pool_impl(static_cast<const float*>(vt), qt, w, w_sb, w_sv, w_sq, out, B, V, Q, D, 1, PoolShift{nullptr, nullptr, 1}, stream, vt_bf16 ? 1 : 0, &qa);
}

extern "C" int cti_bi_logits_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                                 float* logits, int B, int G, int V, int Q, int D, void* stream) {
    CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(h); CTI_REQUIRE_PTR(logits);
    CTI_REQUIRE(B > 0 && G > 0 && V > 0 && Q > 0 && D > 0, CTI_E_SHAPE, "cti_bi_logits_fwd: B=%d G=%d V=%d Q=%d D=%d", B, G, V, Q, D);
    hipLaunchKernelGGL(bi_logits_kernel, dim3((unsigned)(B * V)), dim3(64), 0, as_stream(stream), vt, qt, h, h_scale, h_bias,
                       logits, G, V, Q, D);
    return launch_status("cti_bi_logits_fwd");
}

static int bi_logits_mfma_impl(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                               float* logits, int B, int G, int V, int Q, int D, void* stream, const uint8_t* sm_mask, float* sm_p, int* sm_cnt, int terms = 3, int vt16 = 0);

extern "C" int cti_bi_logits_mfma_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                                      float* logits, int B, int G, int V, int Q, int D, void* stream) {
    return bi_logits_mfma_impl(vt, qt, h, h_scale, h_bias, logits, B, G, V, Q, D, stream, nullptr, nullptr, nullptr);
}

// cti_bi_logits_mfma_fwd in the arithmetic of `prec`: CTI_PREC_BF16 = one bf16 product per pair (the plain-bf16 mode of the model forwards), anything else the
// fp32-grade three-product form.  The single-product form exists for the LDS-staged kernel only (V <= 64, G*Q <= 128, D % 32 == 0): other shapes run fp32-grade.
extern "C" int cti_bi_logits_prec_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                                      float* logits, int B, int G, int V, int Q, int D, int prec, void* stream) {
    return bi_logits_mfma_impl(vt, qt, h, h_scale, h_bias, logits, B, G, V, Q, D, stream, nullptr, nullptr, nullptr, prec == CTI_PREC_BF16 ? 1 : 3);
}

// ... with `vt` as bf16 rows (round 5; the plain-bf16 mode's attention projection written by cti_gemm_bf16_rows): the LDS-staged kernel's shapes only
// (V <= 64, G*Q <= 128, D % 32 == 0), CTI_E_UNSUPPORTED otherwise.
extern "C" int cti_bi_logits_prec_vt16_fwd(const void* vt_bf16, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                                           float* logits, int B, int G, int V, int Q, int D, int prec, void* stream) {
    return bi_logits_mfma_impl(static_cast<const float*>(vt_bf16), qt, h, h_scale, h_bias, logits, B, G, V, Q, D, stream, nullptr, nullptr, nullptr, prec == CTI_PREC_BF16 ? 1 : 3, 1);
}

// BiAttention.forward_all's logits + mask + softmax in ONE launch (reference src/attention.py:29-40 on the projections of src/bc.py:52-57): the bilinear
// logits as cti_bi_logits_mfma_fwd, then the last workgroup of a sample fills the rows of `mask` (B, V; NULL = none) with -inf in `logits` and writes
// p = softmax over (V, Q) per glimpse.  `counters`: B ints that are ZERO at entry and zero again at exit (the caller allocates and zeroes them once).
// CTI_E_UNSUPPORTED (nothing launched) outside V <= 64, Q <= 16, G*Q <= 128, D % 32 == 0, 16-B aligned operands: the caller takes the separate calls.
extern "C" int cti_biattention_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias, const uint8_t* mask,
                                   float* logits, float* p, int* counters, int B, int G, int V, int Q, int D, void* stream) {
    CTI_REQUIRE_PTR(p); CTI_REQUIRE_PTR(counters);
    if (Q > 16) return CTI_E_UNSUPPORTED;
    return bi_logits_mfma_impl(vt, qt, h, h_scale, h_bias, logits, B, G, V, Q, D, stream, mask, p, counters);
}

static int bi_logits_mfma_impl(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                               float* logits, int B, int G, int V, int Q, int D, void* stream, const uint8_t* sm_mask, float* sm_p, int* sm_cnt, int terms, int vt16) {
    CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(h); CTI_REQUIRE_PTR(logits);
    CTI_REQUIRE(B > 0 && B <= 65535 && G > 0 && V > 0 && Q > 0 && D > 0, CTI_E_SHAPE, "cti_bi_logits_mfma_fwd: B=%d G=%d V=%d Q=%d D=%d", B, G, V, Q, D);
    if (D % 16 != 0 || !aligned16(vt) || !aligned16(qt) || !aligned16(h))
        return CTI_E_UNSUPPORTED;                                          // the caller takes cti_bi_logits_fwd (no message: not an error)
#ifndef CTI_BL_LDS
#define CTI_BL_LDS 1
#endif
    const bool lds_form = CTI_BL_LDS && V <= 64 && G * Q <= 128 && D % 32 == 0;
    if (sm_p && !lds_form) return CTI_E_UNSUPPORTED;                 // the fused mask + softmax lives in the LDS form only
    // round 6: the barrier-free K-split form (bi_logits_ks_kernel) where a sample's tiles fit one workgroup: V <= 48, Q <= 16, G <= 8, K long enough to give every wave
    // a few steps, its h slices + the partial outputs within the LDS.  CTI_BL_KS_FORM=0 restores the LDS form (A/B).
    static const bool ks_form_env = [] { const char* e = getenv("CTI_BL_KS_FORM"); return !e || atoi(e) != 0; }();
    if (ks_form_env && !sm_p && V <= 48 && Q <= 16 && G <= 8 && D % 32 == 0 && D >= 512) {
        const int spw = (D / 32 + 7) / 8;
        const size_t lds = std::max((size_t)8 * G * spw * 32 * 4, (size_t)8 * 12 * 4 * 64 * 4);
        if (lds <= 160 * 1024) {
            void (*kern)(const void*, const float*, const float*, const float*, const float*, float*, int, int, int, int, int) =
                terms == 1 ? (vt16 ? bi_logits_ks_kernel<1, 1> : bi_logits_ks_kernel<1, 0>) : (vt16 ? bi_logits_ks_kernel<3, 1> : bi_logits_ks_kernel<3, 0>);
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return fail((int)e, "cti_bi_logits_mfma_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(kern, dim3(B), dim3(512), lds, as_stream(stream), vt, qt, h, h_scale, h_bias, logits, G, V, Q, D, spw);
            return launch_status("cti_bi_logits_mfma_fwd");
        }
    }
    if (lds_form) {                                                  // left operand split once per workgroup (see bi_logits_lds_kernel)
        static const int ks_env = [] { const char* e = getenv("CTI_BL_KS"); return e ? atoi(e) : 0; }();        // (A/B knob: K ranges per sample)
        // TWO ranges at most by default: the partial sums meet by atomicAdd on the zero-filled output, and only a + b is the same bits in either order -- three ranges measured
        // 8 % faster (99 -> 91 us at B = 256, G = 8, D = 3072; 4 -> 126, 8 -> 141) and made the logits differ from run to run (caught by the replay == eager test)
        const int KS = ks_env > 0 ? ks_env : (D >= 1024 ? 2 : 1);
        const int dper = ((D / 32 + KS - 1) / KS) * 32;
        if (KS > 1) { int rcz = zero_fill(logits, (int64_t)B * G * V * Q, as_stream(stream)); if (rcz) return rcz; }
        // column tiles per workgroup: four (one per wave) once that still gives every CU several workgroups' worth of loads in flight
        const int NT = (G * Q + 15) / 16;
#ifndef CTI_BL_NTW
#define CTI_BL_NTW 8        // column tiles per workgroup.  8 = all of them (one workgroup per sample and K half).  Measured at B = 256, G = 8, D = 3072
#endif                      // (profiles/r03_hbm_kernels.jsonl): 8 -> 96 us, 4 -> 135, 2 -> 174, 1 -> 262: every extra workgroup re-splits the vt slice, and that VALU work is the bound
        static const int ntw_max = [] { const char* e = getenv("CTI_BL_NTW"); const int v = e ? atoi(e) : CTI_BL_NTW; return v < 1 ? 1 : (v > 8 ? 8 : v); }();   // (A/B knob; 8 = the round-2 form)
        static const bool ntw_set = getenv("CTI_BL_NTW") != nullptr;
        // (one product per pair: the vt slice is only converted, not split, so a second workgroup per sample and K range pays: 89 -> 81 us at B = 256, G = 8, D = 3072)
        const int ntw_lim = (terms == 1 && !ntw_set) ? 4 : ntw_max;
        const int NTW = NT > ntw_lim ? ntw_lim : NT, NZ = (NT + NTW - 1) / NTW;
        if (terms == 1)
            hipLaunchKernelGGL(bi_logits_lds_kernel<1>, dim3(B, KS, NZ), dim3(256), 0, as_stream(stream), vt, qt, h, h_scale, h_bias, logits, G, V, Q, D, (V + 15) / 16,
                               NT, dper, KS > 1 ? 1 : 0, NTW, sm_mask, sm_p, sm_cnt, KS * NZ, vt16);
        else
            hipLaunchKernelGGL(bi_logits_lds_kernel<3>, dim3(B, KS, NZ), dim3(256), 0, as_stream(stream), vt, qt, h, h_scale, h_bias, logits, G, V, Q, D, (V + 15) / 16,
                               NT, dper, KS > 1 ? 1 : 0, NTW, sm_mask, sm_p, sm_cnt, KS * NZ, vt16);
        return launch_status("cti_bi_logits_mfma_fwd");
    }
    if (vt16) return CTI_E_UNSUPPORTED;                             // bf16 rows: the LDS-staged kernel only
    const int MT = (V + 31) / 32, NT = (G * Q + 31) / 32;
    const int KS = D >= 1024 ? 2 : 1;
    const int dper = ((D / 16 + KS - 1) / KS) * 16;
    int rcz = zero_fill(logits, (int64_t)B * G * V * Q, as_stream(stream)); if (rcz) return rcz;
    hipLaunchKernelGGL(bi_logits_mfma_kernel, dim3(B, KS), dim3(512), 0, as_stream(stream), vt, qt, h, h_scale, h_bias, logits, G, V, Q, D, MT, NT, dper);
    return launch_status("cti_bi_logits_mfma_fwd");
}

static int tri_pool_mfma_impl(const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                              int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, int v_rep, PoolShift sh, void* stream, int terms = 3, int vt16 = 0) {
    CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(at); CTI_REQUIRE_PTR(w); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && A > 0 && D > 0 && B <= 65535, CTI_E_SHAPE, "cti_tri_pool_mfma_fwd: B=%d V=%d Q=%d A=%d D=%d", B, V, Q, A, D);
    CTI_REQUIRE(v_rep >= 1 && B % v_rep == 0, CTI_E_SHAPE, "cti_tri_pool_mfma_fwd: v_rep=%d does not divide B=%d", v_rep, B);
    // measured at B = 256, D = 1024: round 3 (fragments in registers, two workgroups per sample): A = 6 35.4 us here vs 46.5 in the streaming VALU form, A = 3 29.1 here
    // vs 27.1 in the product-table VALU form; round 4 (split slice in LDS, eight waves, one workgroup per sample): A = 6 19.8 us, A = 3 18.8 -- both take this kernel now
#ifndef CTI_TRI_MFMA_A3
#define CTI_TRI_MFMA_A3 1
#endif
    if (!(A == 6 || (A == 3 && CTI_TRI_MFMA_A3)) || Q > 16 || V > 64 || D % 32 != 0) return CTI_E_UNSUPPORTED;   // the caller takes cti_tri_pool_fwd
    const int KS = (Q * A + 15) / 16;
    const int tiles = D / 32;
    static const int tpw_env = [] { const char* e = getenv("CTI_TPM_TPW"); return e ? atoi(e) : 0; }();       // (A/B knob)
    // ONE workgroup per sample where the sample has 32 tiles or more: every workgroup compacts and splits the sample's 64 x QA attention slice, a second workgroup
    // repeats that, and that set-up is what the kernel's time is made of (round 4, A = 6, D = 1 024: 4 waves x 4 tiles = two workgroups per sample 37.6 us, 4 x 8 = one
    // 29.4, 4 x 2 51, 4 x 1 80; eight waves per workgroup do not fit the register file: 212 B of scratch, 37-60 us)
    static const int nw_env = [] { const char* e = getenv("CTI_TPM_NW"); return e ? atoi(e) : 0; }();
    const int nw = (nw_env == 4 || nw_env == 8) ? nw_env : 8;
    const int tpw = tpw_env > 0 ? tpw_env : (tiles >= 32 ? (tiles + nw - 1) / nw : 1);
    const dim3 grid((tiles + nw * tpw - 1) / (nw * tpw), B);
    const size_t lds = sizeof(unsigned short) * 2 * 64 * (size_t)(KS * 16 + 8);
#define CTI_TM1(Av, KSv, Tv, Hv) hipLaunchKernelGGL((tri_pool_mfma_kernel<Av, KSv, Tv, Hv>), grid, dim3(64 * nw), lds, as_stream(stream), vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, V, Q, D, tpw, v_rep, sh)
#define CTI_TM(Av, KSv) { if (terms == 1) { if (vt16) CTI_TM1(Av, KSv, 1, 1); else CTI_TM1(Av, KSv, 1, 0); } \
                          else            { if (vt16) CTI_TM1(Av, KSv, 3, 1); else CTI_TM1(Av, KSv, 3, 0); } }
    if (A == 3) { if (KS <= 2) CTI_TM(3, 2) else CTI_TM(3, 3) }
    else        { if (KS <= 4) CTI_TM(6, 4) else if (KS == 5) CTI_TM(6, 5) else CTI_TM(6, 6) }
#undef CTI_TM
    return launch_status("cti_tri_pool_mfma_fwd");
}

extern "C" int cti_tri_pool_mfma_fwd(const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                                     int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, int v_rep, void* stream) {
    return tri_pool_mfma_impl(vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, B, V, Q, A, D, v_rep, PoolShift{nullptr, nullptr, 0}, stream);
}

// out[b,d] = sum_vqa vt[b / v_rep, v, d] w[b,v,q,a] relu(qt[b,q,d] + qadd[b,d]) relu(at[b,a,d] + aadd[b,d])   (qadd / aadd may be NULL = 0).
// use_mfma: 1 = the fp32-grade MFMA form (three bf16 products per pair) where it applies (A = 3 or 6), 2 = the same with ONE product per pair (the plain-bf16 mode),
// 0 or no such kernel for the shape = the product-table / streaming VALU kernels (v_rep must then be 1).
// CTI_E_UNSUPPORTED when no kernel with the on-load shift takes the shape: nothing is launched, the caller materialises the operands.
extern "C" int cti_tri_pool_shift_fwd(const float* vt, const float* qt, const float* at, const float* qadd, const float* aadd, const float* w,
                                      int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, int v_rep,
                                      int use_mfma, void* stream) {
    const PoolShift sh{qadd, aadd, 1};
    if (use_mfma) {
        const int rc = tri_pool_mfma_impl(vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, B, V, Q, A, D, v_rep, sh, stream, use_mfma == 2 ? 1 : 3);
        if (rc != CTI_E_UNSUPPORTED) return rc;
    }
    if (v_rep != 1) return CTI_E_UNSUPPORTED;
    return tri_pool_impl(vt, qt, at, w, w_sb, w_sv, w_sq, w_sa, out, B, V, Q, A, D, sh, stream);
}

// ... with `vt` as bf16 rows (round 5): the MFMA form only (A = 3 or 6); CTI_E_UNSUPPORTED otherwise -- the caller widens the rows.
extern "C" int cti_tri_pool_shift_vt16_fwd(const void* vt_bf16, const float* qt, const float* at, const float* qadd, const float* aadd, const float* w,
                                           int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, int v_rep,
                                           int use_mfma, void* stream) {
    if (!use_mfma) return CTI_E_UNSUPPORTED;
    return tri_pool_mfma_impl(static_cast<const float*>(vt_bf16), qt, at, w, w_sb, w_sv, w_sq, w_sa, out, B, V, Q, A, D, v_rep, PoolShift{qadd, aadd, 1}, stream,
                              use_mfma == 2 ? 1 : 3, 1);
}

extern "C" int cti_pool_dw_mfma(const float* dout, const float* vt, const float* qt, const float* at, float* dw, int B, int V, int Q, int A, int D,
                                void* stream) {
    CTI_REQUIRE_PTR(dout); CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(dw);
    CTI_REQUIRE(B > 0 && B <= 65535 && V > 0 && Q > 0 && A > 0 && D > 0, CTI_E_SHAPE, "cti_pool_dw_mfma: B=%d V=%d Q=%d A=%d D=%d", B, V, Q, A, D);
    if (D % 16 != 0 || !aligned16(dout) || !aligned16(vt) || !aligned16(qt) || (at && !aligned16(at))) return CTI_E_UNSUPPORTED;
    const int MT = (V + 31) / 32, NT = (Q * A + 31) / 32;
    const int KS = D >= 512 ? 2 : 1;                               // at most TWO addends meet by atomicAdd on the zero-filled output: a + b in either order, the same bits
    const int dper = ((D / 16 + KS - 1) / KS) * 16;
    int rcz = zero_fill(dw, (int64_t)B * V * Q * A, as_stream(stream)); if (rcz) return rcz;
    hipLaunchKernelGGL(pool_dw_mfma_kernel, dim3(B, KS), dim3(256), 0, as_stream(stream), dout, vt, qt, at, dw, V, Q, A, D, MT, NT, dper);
    return launch_status("cti_pool_dw_mfma");
}

extern "C" int cti_bi_logits_bwd_mfma(const float* dlogits, const float* vt, const float* qt, const float* h, const float* h_scale, float* dvt,
                                      float* dqt, float* dh_partial, int B, int G, int V, int Q, int D, void* stream) {
    CTI_REQUIRE_PTR(dlogits); CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(h); CTI_REQUIRE_PTR(dvt); CTI_REQUIRE_PTR(dqt); CTI_REQUIRE_PTR(dh_partial);
    CTI_REQUIRE(B > 0 && B <= 65535 && G > 0 && V > 0 && Q > 0 && D > 0, CTI_E_SHAPE, "cti_bi_logits_bwd_mfma: B=%d G=%d V=%d Q=%d D=%d", B, G, V, Q, D);
    if (G > BLG || V > 64 || Q > 16 || D % 32 != 0) return CTI_E_UNSUPPORTED;       // the caller takes cti_bi_logits_bwd
    const int tiles = D / 32;
    const int tpw = tiles >= 64 ? 4 : (tiles >= 16 ? 2 : 1);
    const dim3 grid((tiles + 4 * tpw - 1) / (4 * tpw), B);
    const size_t lds1 = sizeof(float) * (size_t)G * 64 * 20, lds2 = sizeof(float) * (size_t)G * 32 * 68;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bi_logits_bwd_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail((int)e, "cti_bi_logits_bwd_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    hipLaunchKernelGGL(bi_logits_bwd_vh_kernel, grid, dim3(256), lds1, as_stream(stream), dlogits, vt, qt, h, h_scale, dvt, dh_partial, G, V, Q, D, tpw);
    int rc = launch_status("cti_bi_logits_bwd_mfma/dvt"); if (rc) return rc;
    hipLaunchKernelGGL(bi_logits_bwd_q_kernel, grid, dim3(256), lds2, as_stream(stream), dlogits, vt, h, h_scale, dqt, G, V, Q, D, tpw);
    return launch_status("cti_bi_logits_bwd_mfma/dqt");
}
