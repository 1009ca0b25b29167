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

}  // namespace
}  // namespace cti

using namespace cti;

// chunking of the Tri softmax: chunks of <= 32768 positions (x G floats), at least one per sample
static void tri_chunks(int V, int64_t QA, int64_t* chunk_n, int* nchunk) {
    const int64_t N = (int64_t)V * QA;
    const int64_t c = 32768;
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
    hipLaunchKernelGGL(tri_normalise_kernel, dim3(gx, B), dim3(256), 0, st, logits, stats, p, NG, G);
    return launch_status("cti_masked_softmax_tri_fwd/normalise");
}

extern "C" int cti_masked_softmax_bi_fwd(float* logits, const uint8_t* mask, float* p, int B, int G, int V, int Q, void* stream) {
    CTI_REQUIRE_PTR(logits); CTI_REQUIRE_PTR(p);
    CTI_REQUIRE(B > 0 && G > 0 && V > 0 && Q > 0, CTI_E_SHAPE, "cti_masked_softmax_bi_fwd: B=%d G=%d V=%d Q=%d", B, G, V, Q);
    const int rows = B * G;
    hipLaunchKernelGGL(bi_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), logits, mask, p, rows, G, V, Q);
    return launch_status("cti_masked_softmax_bi_fwd");
}

extern "C" int cti_tri_pool_fwd(const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                                int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, void* stream) {
    CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(at); CTI_REQUIRE_PTR(w); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && A > 0 && D > 0 && B <= 65535, CTI_E_SHAPE, "cti_tri_pool_fwd: B=%d V=%d Q=%d A=%d D=%d", B, V, Q, A, D);
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

extern "C" int cti_bi_pool_fwd(const float* vt, const float* qt, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                               float* out, int B, int V, int Q, int D, int k, void* stream) {
    CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && D > 0 && k > 0 && B <= 65535, CTI_E_SHAPE, "cti_bi_pool_fwd: B=%d V=%d Q=%d D=%d k=%d", B, V, Q, D, k);
    CTI_REQUIRE(D >= k, CTI_E_SHAPE, "cti_bi_pool_fwd: D=%d < k=%d", D, k);
    const size_t lds = sizeof(float) * 256 * (size_t)Q * k;
    CTI_REQUIRE(lds <= 64 * 1024, CTI_E_SHAPE, "cti_bi_pool_fwd: Q*k=%d too large for the LDS staging (<= 64)", Q * k);
    const int NO = D / k;                                   // AvgPool1d(k, stride=k) drops a ragged tail, like torch
    hipLaunchKernelGGL(bi_pool_kernel, dim3((NO + 255) / 256, B), dim3(256), lds, as_stream(stream), vt, qt, w, w_sb, w_sv, w_sq,
                       out, V, Q, D, k);
    return launch_status("cti_bi_pool_fwd");
}

extern "C" int cti_bi_logits_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                                 float* logits, int B, int G, int V, int Q, int D, void* stream) {
    CTI_REQUIRE_PTR(vt); CTI_REQUIRE_PTR(qt); CTI_REQUIRE_PTR(h); CTI_REQUIRE_PTR(logits);
    CTI_REQUIRE(B > 0 && G > 0 && V > 0 && Q > 0 && D > 0, CTI_E_SHAPE, "cti_bi_logits_fwd: B=%d G=%d V=%d Q=%d D=%d", B, G, V, Q, D);
    hipLaunchKernelGGL(bi_logits_kernel, dim3((unsigned)(B * V)), dim3(64), 0, as_stream(stream), vt, qt, h, h_scale, h_bias,
                       logits, G, V, Q, D);
    return launch_status("cti_bi_logits_fwd");
}
