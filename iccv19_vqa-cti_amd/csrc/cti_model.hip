// cti_model.hip -- the rows either side of the CTI path: word-embedding gather / scatter (the GRU is
// cti_gru.hip), Swish, the residual broadcast-add / sequence sums of the model forwards, and
// the two losses.  All of it is HBM-bound elementwise or row-reduction work: coalesced accesses, one pass over the data.
#include "cti_common.h"
#include <cstdlib>

namespace cti {
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// out[i, 0:dim] = table0[tok[i]], out[i, dim:2*dim] = table1[tok[i]] (when table1).  A token outside [0, rows) yields NaN.
__global__ void embedding_fwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ t0, const float* __restrict__ t1,
                                     float* __restrict__ out, int64_t n, int dim, int64_t rows) {
    const int width = t1 ? 2 * dim : dim;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t k = tok[i];
        const bool ok = k >= 0 && k < rows;
        for (int c = threadIdx.x; c < width; c += blockDim.x) {
            const float* src = c < dim ? t0 + k * dim + c : t1 + k * dim + (c - dim);
            out[i * width + c] = ok ? *src : __builtin_nanf("");
        }
    }
}

// The same lookup as bf16 rows of pitch ld (>= width, zero-filled beyond it): the A operand of the GRU's input-side product as it stands (round 6: no fp32 rows, no split
// pass in front of that product in the plain-bf16 mode)
__global__ void embedding_fwd_bf16_kernel(const int64_t* __restrict__ tok, const float* __restrict__ t0, const float* __restrict__ t1,
                                          unsigned short* __restrict__ out, int64_t ld, int64_t n, int dim, int64_t rows) {
    const int width = t1 ? 2 * dim : dim;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t k = tok[i];
        const bool ok = k >= 0 && k < rows;
        for (int c = threadIdx.x; c < (int)ld; c += blockDim.x) {
            float x = 0.f;
            if (c < width) x = ok ? (c < dim ? t0[k * dim + c] : t1[k * dim + (c - dim)]) : __builtin_nanf("");
            out[i * ld + c] = __builtin_bit_cast(unsigned short, static_cast<__bf16>(x));
        }
    }
}

// dtable[tok[i], c] += dout[i, col_off + c]; the padding row receives nothing (nn.Embedding(padding_idx)).
// DETERMINISTIC: no atomics.  The workgroup of position i scans the token list in chunks of 256 (one ballot per wave, the four masks shared
// through LDS); if the first occurrence of its token is not i it has nothing to do, otherwise it alone owns table row tok[i] and adds the
// gradient rows of every occurrence in ascending position order: dtable[k] += ((g_j1 + g_j2) + g_j3) + ...  -- the same bits on every run
// and for every grid size (a few thousand tokens per step: the scans cost ~14 chunk iterations per workgroup).
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ dout, int64_t ld, int col_off,
                                                            float* __restrict__ dtable, int64_t n, int dim, int64_t rows, int64_t pad) {
    __shared__ unsigned long long masks[2][4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t k = tok[i];
        if (k < 0 || k >= rows || k == pad) continue;                // (block-uniform)
        float acc[4] = {0.f, 0.f, 0.f, 0.f};                         // columns t, t + 256, ... (dim <= 1024 takes one pass; wider tables loop below)
        for (int c0 = 0; c0 < dim; c0 += 1024) {
            bool first_seen = false, mine = true;
            int buf = 0;
            for (int64_t j0 = 0; j0 < n && mine; j0 += 256, buf ^= 1) {
                const int64_t j = j0 + t;
                const unsigned long long m = __ballot(j < n && tok[j] == k);
                if (lane == 0) masks[buf][w] = m;
                __syncthreads();                                     // (two buffers: the next chunk's writes cannot overtake this chunk's readers)
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    unsigned long long mm = masks[buf][ww];
                    while (mm) {
                        const int64_t jj = j0 + ww * 64 + __builtin_ctzll(mm);
                        mm &= mm - 1;
                        if (!first_seen) { first_seen = true; if (jj != i) { mine = false; break; } }
                        const float* g = dout + jj * ld + col_off + c0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (c0 + t + u * 256 < dim) acc[u] += g[t + u * 256];
                    }
                    if (!mine) break;
                }
            }
            if (mine) {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (c0 + t + u * 256 < dim) { dtable[k * dim + c0 + t + u * 256] += acc[u]; acc[u] = 0.f; }
            } else {
                break;
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

__global__ void swish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * sigmoidf_(x[i]);
}
__global__ void swish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float s = sigmoidf_(x[i]); dx[i] = dy[i] * (s + x[i] * s * (1.f - s)); }
}

// out[b,h] = beta * out[b,h] + sum_l x[b,l,h]
__global__ void seq_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int L, int H, float beta) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * H) return;
    const int b = (int)(idx / H), h = (int)(idx % H);
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += x[((int64_t)b * L + l) * H + h];
    out[idx] = (beta != 0.f ? beta * out[idx] : 0.f) + s;
}
// eq[b] = 1 when rows b and b - 1 of x (row_bytes each, a multiple of 16) hold the same BITS, 0 otherwise (eq[0] = 0): the MC pipeline's repeated images
// (src/MC/train.py:75-79 feeds every image once per candidate answer).  One workgroup per row; a mismatch anywhere clears the byte.
// Round 5: 1 024 threads per row with eight independent 16-B loads in flight per thread and round, no data-dependent exit inside a round (the first form's 256
// threads walked their 72 pieces one dependent load pair at a time: 44.6 us for 37 MB, latency-bound -- 3.8 % of the MC model's forward).
__global__ __launch_bounds__(1024) void rows_equal_prev_kernel(const uint4* __restrict__ x, int64_t row_vec, unsigned char* __restrict__ eq) {
    const int b = blockIdx.x;
    __shared__ int diff;
    if (threadIdx.x == 0) diff = 0;
    __syncthreads();
    if (b > 0) {
        const uint4* p = x + (int64_t)b * row_vec;
        const uint4* q = p - row_vec;
        unsigned d = 0;
        constexpr int U = 4;
        for (int64_t i0 = threadIdx.x; i0 < row_vec && !d; i0 += 1024 * U) {
            uint4 u[U], w[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int64_t i = i0 + (int64_t)k * 1024;
                const int64_t ic = i < row_vec ? i : i0;                       // (a short tail re-reads the round's first piece)
                u[k] = p[ic]; w[k] = q[ic];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) d |= (u[k].x ^ w[k].x) | (u[k].y ^ w[k].y) | (u[k].z ^ w[k].z) | (u[k].w ^ w[k].w);
        }
        if (d) diff = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) eq[b] = (b > 0 && !diff) ? 1 : 0;
}
// out[i] = NaN for every i unless eq says that the batch is made of groups of r identical rows (eq[b] == 1 for every b with b % r != 0): the check behind
// TanModel.v_replication = 'auto' -- a batch that breaks the assumption the forward was run under gives NaNs, never a plausible wrong answer
__global__ __launch_bounds__(256) void poison_unless_replicated_kernel(const unsigned char* __restrict__ eq, int B, int r, float* __restrict__ out, int64_t n) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 256) if (b % r != 0 && !eq[b]) bad = 1;
    __syncthreads();
    if (!bad) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = __builtin_nanf("");
}
// y[r, n] = act(scale * sum_k x[r, k] W[n, k] + bias[n]) for a HANDFUL of outputs (N <= 8: the MC models' answer head, src/classifier.py:26 with out_dim = 2) in exact fp32:
// one workgroup per row, its threads stride over K with N running sums, one block reduction.  The GEMM path costs this 256 x 2 x 2 048 product a split-K launch pair.
template <int N>
__global__ __launch_bounds__(256) void linear_small_n_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int64_t ldw, const float* __restrict__ scale,
                                                             const float* __restrict__ bias, float* __restrict__ y, int64_t ldy, int K, int relu) {
    const int r = blockIdx.x, t = threadIdx.x;
    const float* xr = x + (int64_t)r * ldx;
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    for (int k = t; k < K; k += 256) {
        const float xv = xr[k];
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = fmaf(xv, W[(int64_t)n * ldw + k], acc[n]);
    }
    __shared__ float red[N][4];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float v = acc[n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((t & 63) == 0) red[n][t >> 6] = v;
    }
    __syncthreads();
    if (t < N) {
        float v = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
        v = v * (scale ? scale[0] : 1.f) + (bias ? bias[t] : 0.f);
        y[(int64_t)r * ldy + t] = relu ? fmaxf(v, 0.f) : v;
    }
}
// out[b,h] = cq * sum_l q[b,l,h] + ca * sum_l a[b,l,h] + dq * Dq[b,h] + da * Da[b,h]   (a / Dq / Da may be NULL): the classifier input of the hoisted glimpse
// loops in ONE pass -- CTI: q_emb_0.sum(1) + ans_emb_0.sum(1) + Lq Dq + La Da (src/FFOE/base_model.py:134); BAN: G q_emb_0.sum(1) + L sum_g D_g (:63-64)
__global__ void joint_sums_kernel(const float* __restrict__ q, int Lq, float cq, const float* __restrict__ a, int La, float ca, const float* __restrict__ Dq, float dq,
                                  const float* __restrict__ Da, float da, float* __restrict__ out, int B, int H) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * H) return;
    const int b = (int)(idx / H), h = (int)(idx % H);
    float sq = 0.f, sa = 0.f;
    for (int l = 0; l < Lq; ++l) sq += q[((int64_t)b * Lq + l) * H + h];
    if (a) for (int l = 0; l < La; ++l) sa += a[((int64_t)b * La + l) * H + h];
    float v = cq * sq + ca * sa;
    if (Dq) v += dq * Dq[idx];
    if (Da) v += da * Da[idx];
    out[idx] = v;
}
// out = a x + b y  (the classifier input of the hoisted glimpse loops: G * q_emb_0.sum(1) + L * sum_g D_g)
__global__ void axpby_kernel(const float* __restrict__ x, float a, const float* __restrict__ y, float b, float* __restrict__ out, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = a * x[idx] + b * y[idx];
}
// out[b,l,h] = (x ? x[b,l,h] : 0) + y[b,h]
__global__ void seq_bcast_add_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int B, int L, int H) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * L * H) return;
    const int h = (int)(idx % H);
    const int64_t b = idx / ((int64_t)L * H);
    out[idx] = (x ? x[idx] : 0.f) + y[b * H + h];
}

// The residual projection of the model forwards in one pass behind the GEMM (reference src/FFOE/base_model.py:61,131-132,134):
//   y[b, n]      = scale[n / div] * sum_s part[s][b][n] + bias[n]           (split-K partials of x @ W^T; S = 1: the plain product)
//   out[b, l, n] = seq[b, l, n] + y[b, n]                                  (q_prj(b_emb.unsqueeze(1)) + q_emb)
//   acc[b, n]    = beta * acc[b, n] + sum_l out[b, l, n]                   (optional: the q_emb.sum(1) the classifier input is built from)
// One float4 of a (b, n) column per thread: the S partials and the L sequence rows are coalesced across the workgroup.  Replaces the
// split-K reduce kernel + the broadcast-add kernel + the sequence-sum kernel of a glimpse (three launches, two extra passes over seq).
__global__ __launch_bounds__(256) void linear_residual_kernel(const float* __restrict__ part, int S, const float* __restrict__ scale, int scale_div,
                                                               const float* __restrict__ bias, const float* __restrict__ seq, float* __restrict__ out,
                                                               float* __restrict__ acc, float beta, int B, int L, int N) {
    const int n4 = N >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * n4) return;
    const int b = (int)(idx / n4), n0 = (int)(idx % n4) * 4;
    const int64_t BN = (int64_t)B * N;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
        const float4 x = *reinterpret_cast<const float4*>(part + s * BN + (int64_t)b * N + n0);
        y.x += x.x; y.y += x.y; y.z += x.z; y.w += x.w;
    }
    const float4 bi = bias ? *reinterpret_cast<const float4*>(bias + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
    y.x = y.x * (scale ? scale[n0 / scale_div] : 1.f) + bi.x;             y.y = y.y * (scale ? scale[(n0 + 1) / scale_div] : 1.f) + bi.y;
    y.z = y.z * (scale ? scale[(n0 + 2) / scale_div] : 1.f) + bi.z;       y.w = y.w * (scale ? scale[(n0 + 3) / scale_div] : 1.f) + bi.w;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* sp = seq + ((int64_t)b * L) * N + n0;
    float* op = out + ((int64_t)b * L) * N + n0;
    for (int l = 0; l < L; ++l) {
        float4 v = *reinterpret_cast<const float4*>(sp + (int64_t)l * N);
        v.x += y.x; v.y += y.y; v.z += y.z; v.w += y.w;
        *reinterpret_cast<float4*>(op + (int64_t)l * N) = v;
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    if (acc) {
        float4* ap = reinterpret_cast<float4*>(acc + (int64_t)b * N + n0);
        if (beta != 0.f) { const float4 o = *ap; sum.x += beta * o.x; sum.y += beta * o.y; sum.z += beta * o.z; sum.w += beta * o.w; }
        *ap = sum;
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
    for (int i = 1; i < nw; ++i) s = fmaxf(s, red[i]);
    return s;
}

// row_loss[r] = sum_c max(x,0) - x*t + log1p(exp(-|x|))   (BCEWithLogitsLoss, reduction = sum, one row per workgroup)
__global__ void bce_rows_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ row_loss, int n) {
    __shared__ float red[16];
    const int64_t r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const float v = x[r * n + c];
        s += fmaxf(v, 0.f) - v * t[r * n + c] + log1pf(__expf(-fabsf(v)));
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) row_loss[r] = s;
}
// dx = beta*dx + (sigmoid(x) - t) * coef * (*up)
__global__ void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ up, float coef,
                               float* __restrict__ dx, int64_t n, float beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = coef * (up ? *up : 1.f);
    dx[i] = (beta != 0.f ? beta * dx[i] : 0.f) + (sigmoidf_(x[i]) - t[i]) * c;
}

// row_kl[r] = sum_c pk * (log pk - log ps),  pk = softmax(k / T), ps = softmax(x / T)
__global__ void kd_rows_fwd_kernel(const float* __restrict__ x, const float* __restrict__ k, float* __restrict__ row_kl, int n, float invT) {
    __shared__ float red[16];
    const int64_t r = blockIdx.x;
    const float* xr = x + r * n; const float* kr = k + r * n;
    float mx = -INFINITY, mk = -INFINITY;
    for (int c = threadIdx.x; c < n; c += blockDim.x) { mx = fmaxf(mx, xr[c] * invT); mk = fmaxf(mk, kr[c] * invT); }
    mx = block_max(mx, red); mk = block_max(mk, red);
    float sx = 0.f, sk = 0.f;
    for (int c = threadIdx.x; c < n; c += blockDim.x) { sx += __expf(xr[c] * invT - mx); sk += __expf(kr[c] * invT - mk); }
    sx = block_sum(sx, red); sk = block_sum(sk, red);
    const float lsx = mx + __logf(sx), lsk = mk + __logf(sk);
    float acc = 0.f;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const float lk = kr[c] * invT - lsk, ls = xr[c] * invT - lsx;
        acc += __expf(lk) * (lk - ls);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) row_kl[r] = acc;
}
// dx[r,c] = beta*dx + coef * (*up) * invT * (softmax(x/T) - softmax(k/T))
__global__ void kd_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ k, const float* __restrict__ up, float coef,
                                   float* __restrict__ dx, int n, float invT, float beta) {
    __shared__ float red[16];
    const int64_t r = blockIdx.x;
    const float* xr = x + r * n; const float* kr = k + r * n;
    float mx = -INFINITY, mk = -INFINITY;
    for (int c = threadIdx.x; c < n; c += blockDim.x) { mx = fmaxf(mx, xr[c] * invT); mk = fmaxf(mk, kr[c] * invT); }
    mx = block_max(mx, red); mk = block_max(mk, red);
    float sx = 0.f, sk = 0.f;
    for (int c = threadIdx.x; c < n; c += blockDim.x) { sx += __expf(xr[c] * invT - mx); sk += __expf(kr[c] * invT - mk); }
    sx = block_sum(sx, red); sk = block_sum(sk, red);
    const float cc = coef * (up ? *up : 1.f) * invT;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        const float g = cc * (__expf(xr[c] * invT - mx) / sx - __expf(kr[c] * invT - mk) / sk);
        dx[r * n + c] = (beta != 0.f ? beta * dx[r * n + c] : 0.f) + g;
    }
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace
}  // namespace cti

using namespace cti;

extern "C" {

int cti_embedding_fwd(const int64_t* tokens, const float* table0, const float* table1, float* out, int64_t n, int dim, int64_t rows,
                      void* stream) {
    CTI_REQUIRE_PTR(tokens); CTI_REQUIRE_PTR(table0); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(n >= 0 && dim > 0 && rows > 0, CTI_E_SHAPE, "cti_embedding_fwd: n=%lld dim=%d rows=%lld", (long long)n, dim, (long long)rows);
    if (n == 0) return CTI_OK;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)(n < 65535 ? n : 65535)), dim3(256), 0, as_stream(stream), tokens, table0,
                       table1, out, n, dim, rows);
    return launch_status("cti_embedding_fwd");
}

int cti_embedding_fwd_bf16(const int64_t* tokens, const float* table0, const float* table1, void* out_bf16, int64_t ld_out, int64_t n, int dim, int64_t rows,
                           void* stream) {
    CTI_REQUIRE_PTR(tokens); CTI_REQUIRE_PTR(table0); CTI_REQUIRE_PTR(out_bf16);
    CTI_REQUIRE(n >= 0 && dim > 0 && rows > 0 && ld_out >= (table1 ? 2 : 1) * (int64_t)dim && ld_out < (1 << 20), CTI_E_SHAPE,
                "cti_embedding_fwd_bf16: n=%lld dim=%d rows=%lld ld=%lld", (long long)n, dim, (long long)rows, (long long)ld_out);
    if (n == 0) return CTI_OK;
    hipLaunchKernelGGL(embedding_fwd_bf16_kernel, dim3((unsigned)(n < 65535 ? n : 65535)), dim3(256), 0, as_stream(stream), tokens, table0, table1,
                       static_cast<unsigned short*>(out_bf16), ld_out, n, dim, rows);
    return launch_status("cti_embedding_fwd_bf16");
}

int cti_embedding_bwd(const int64_t* tokens, const float* dout, int64_t ld_dout, int col_off, float* dtable, int64_t n, int dim,
                      int64_t rows, int64_t padding_idx, void* stream) {
    CTI_REQUIRE_PTR(tokens); CTI_REQUIRE_PTR(dout); CTI_REQUIRE_PTR(dtable);
    CTI_REQUIRE(n >= 0 && dim > 0 && rows > 0 && col_off >= 0 && ld_dout >= col_off + dim, CTI_E_SHAPE,
                "cti_embedding_bwd: n=%lld dim=%d rows=%lld col_off=%d ld=%lld", (long long)n, dim, (long long)rows, col_off, (long long)ld_dout);
    if (n == 0) return CTI_OK;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)(n < 65535 ? n : 65535)), dim3(256), 0, as_stream(stream), tokens, dout,
                       ld_dout, col_off, dtable, n, dim, rows, padding_idx);
    return launch_status("cti_embedding_bwd");
}

int cti_swish_fwd(const float* x, float* y, int64_t n, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(y);
    if (n <= 0) return CTI_OK;
    hipLaunchKernelGGL(swish_fwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, y, n);
    return launch_status("cti_swish_fwd");
}
int cti_swish_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(dy); CTI_REQUIRE_PTR(dx);
    if (n <= 0) return CTI_OK;
    hipLaunchKernelGGL(swish_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, dy, dx, n);
    return launch_status("cti_swish_bwd");
}

int cti_seq_sum(const float* x, float* out, int B, int L, int H, float beta, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B >= 0 && L >= 0 && H > 0, CTI_E_SHAPE, "cti_seq_sum: B=%d L=%d H=%d", B, L, H);
    if (B == 0) return CTI_OK;
    hipLaunchKernelGGL(seq_sum_kernel, dim3(blocks_for((int64_t)B * H, 256)), dim3(256), 0, as_stream(stream), x, out, B, L, H, beta);
    return launch_status("cti_seq_sum");
}
int cti_rows_equal_prev(const void* x, int64_t row_bytes, int B, unsigned char* eq, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(eq);
    CTI_REQUIRE(B > 0 && B <= 65535 * 64 && row_bytes > 0, CTI_E_SHAPE, "cti_rows_equal_prev: B=%d row_bytes=%lld", B, (long long)row_bytes);
    if (row_bytes % 16 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return CTI_E_UNSUPPORTED;
    hipLaunchKernelGGL(rows_equal_prev_kernel, dim3((unsigned)B), dim3(1024), 0, as_stream(stream), static_cast<const uint4*>(x), row_bytes / 16, eq);
    return launch_status("cti_rows_equal_prev");
}
int cti_poison_unless_replicated(const unsigned char* eq, int B, int r, float* out, int64_t n, void* stream) {
    CTI_REQUIRE_PTR(eq); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B > 0 && r >= 1 && B % r == 0 && n >= 0, CTI_E_SHAPE, "cti_poison_unless_replicated: B=%d r=%d n=%lld", B, r, (long long)n);
    if (n == 0 || r == 1) return CTI_OK;
    hipLaunchKernelGGL(poison_unless_replicated_kernel, dim3((unsigned)(n >= 256 * 64 ? 64 : (n + 255) / 256)), dim3(256), 0, as_stream(stream), eq, B, r, out, n);
    return launch_status("cti_poison_unless_replicated");
}
int cti_linear_small_n(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* scale, const float* bias, float* y, int64_t ldy, int rows, int K, int N,
                       int relu, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(W); CTI_REQUIRE_PTR(y);
    CTI_REQUIRE(rows > 0 && K > 0 && N > 0 && ldx >= K && ldw >= K && ldy >= N, CTI_E_SHAPE, "cti_linear_small_n: rows=%d K=%d N=%d", rows, K, N);
    if (N > 8) return CTI_E_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
#define CTI_LS(Nv) hipLaunchKernelGGL(linear_small_n_kernel<Nv>, dim3((unsigned)rows), dim3(256), 0, st, x, ldx, W, ldw, scale, bias, y, ldy, K, relu)
    switch (N) { case 1: CTI_LS(1); break; case 2: CTI_LS(2); break; case 3: CTI_LS(3); break; case 4: CTI_LS(4); break;
                 case 5: CTI_LS(5); break; case 6: CTI_LS(6); break; case 7: CTI_LS(7); break; default: CTI_LS(8); break; }
#undef CTI_LS
    return launch_status("cti_linear_small_n");
}
int cti_joint_sums(const float* q, int Lq, float cq, const float* a, int La, float ca, const float* Dq, float dq, const float* Da, float da, float* out, int B, int H,
                   void* stream) {
    CTI_REQUIRE_PTR(q); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B > 0 && H > 0 && Lq > 0 && (!a || La > 0), CTI_E_SHAPE, "cti_joint_sums: B=%d H=%d Lq=%d La=%d", B, H, Lq, La);
    hipLaunchKernelGGL(joint_sums_kernel, dim3(blocks_for((int64_t)B * H, 256)), dim3(256), 0, as_stream(stream), q, Lq, cq, a, La, ca, Dq, dq, Da, da, out, B, H);
    return launch_status("cti_joint_sums");
}
int cti_axpby(const float* x, float a, const float* y, float b, float* out, int64_t n, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(y); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(n >= 0, CTI_E_SHAPE, "cti_axpby: n=%lld", (long long)n);
    if (n == 0) return CTI_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, a, y, b, out, n);
    return launch_status("cti_axpby");
}
size_t cti_linear_residual_workspace_bytes(int B, int N, int K, int prec) {
    if (B <= 0 || N <= 0 || K <= 0 || (prec != CTI_PREC_BF16X3 && prec != CTI_PREC_BF16)) return 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int S = plan_ksplit(B, N, planes_kp(K), 1);
    return al(planes_bytes((int64_t)B + PLANE_SLACK_ROWS, K)) + al(sizeof(float) * (size_t)(S > 1 ? S : 1) * (size_t)B * (size_t)N);
}
int cti_linear_residual_pb(const float* x, int64_t ldx, const void* W_planes, const float* scale, int scale_div, const float* bias, const float* seq,
                           float* out, float* acc, float beta, int B, int L, int N, int K, int prec, void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(W_planes); CTI_REQUIRE_PTR(seq); CTI_REQUIRE_PTR(out); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(B > 0 && L > 0 && N > 0 && K > 0 && ldx >= K, CTI_E_SHAPE, "cti_linear_residual_pb: B=%d L=%d N=%d K=%d ldx=%lld", B, L, N, K, (long long)ldx);
    CTI_REQUIRE(N % 4 == 0 && ((reinterpret_cast<uintptr_t>(seq) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(acc) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0,
                CTI_E_ALIGN, "cti_linear_residual_pb: N %% 4 == 0 and 16-B aligned seq / out / acc / bias are required (N=%d)", N);
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_linear_residual_pb: prec=%d (resident planes exist in the bf16 modes only)", prec);
    CTI_REQUIRE(scale == nullptr || scale_div > 0, CTI_E_SHAPE, "cti_linear_residual_pb: scale_div=%d", scale_div);
    const size_t need = cti_linear_residual_workspace_bytes(B, N, K, prec);
    CTI_REQUIRE(workspace_bytes >= need, CTI_E_WORKSPACE, "cti_linear_residual_pb: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    const int Kp = planes_kp(K);
    const int64_t ra = (int64_t)B + PLANE_SLACK_ROWS, rb = (int64_t)N + PLANE_SLACK_ROWS;
    unsigned short* ah = static_cast<unsigned short*>(workspace);
    unsigned short* al_ = ah + (size_t)ra * Kp;
    float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + ((planes_bytes(ra, K) + 255) & ~(size_t)255));
    const unsigned short* bh = static_cast<const unsigned short*>(W_planes);
    const unsigned short* bl = bh + (size_t)rb * Kp;
    // the (B, K) activation is read as fp32 rows by the product itself (no split launch) when its rows are 16-B aligned
    static const bool af32_ok = [] { const char* e = getenv("CTI_AF32_PB"); return e && e[0] == '1'; }();       // (experiment, measured slower: cti_backward.hip af32_pb)
    static const int af32_rows = [] { const char* e = getenv("CTI_AF32_ROWS"); return e ? atoi(e) : 256; }();   // (as cti_gemm_nt_pb: batch-sized products read their fp32 rows themselves)
    const bool af32 = (af32_ok || B <= af32_rows) && (K & 3) == 0 && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    int rc = CTI_OK;
    if (!af32) { rc = split_planes(x, ldx, B, K, ah, prec == CTI_PREC_BF16 ? nullptr : al_, ra, st); if (rc) return rc; }
    PlaneGemmArgs g{};
    if (af32) { g.Af = x; g.ldaf = ldx; g.Kreal = K; }
    g.Ah = ah; g.Al = al_; g.Bh = bh; g.Bl = bl; g.rows_allocA = ra; g.rows_allocB = rb; g.nb1 = 1; g.nb2 = 1;
    g.M = B; g.N = N; g.Kp = Kp; g.terms = prec == CTI_PREC_BF16X3 ? 3 : 1; g.epi = 0; g.gdiv = 1; g.scale_div = 1;
    int S = plan_ksplit(B, N, Kp, 1);
    g.C = part; g.ldc_m = N; g.ldc_n = 1;                        // S == 1: the plain product lands where the partials would
    if (S > 1 && tuning_gemm_cfg() < 0 && gemm_skinny_eligible(g)) { S = 1; rc = gemm_skinny(g, st); if (rc) return rc; }      // round 6: ONE launch, ONE slab (cti_gemm_skinny.hip)
    else {
        if (S > 1) { g.ksplit = S; g.partial = part; g.partials_only = 1; }
        rc = gemm_nt_planes(g, st); if (rc) return rc;
    }
    const int64_t items = (int64_t)B * (N / 4);
    hipLaunchKernelGGL(linear_residual_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, part, S > 1 ? S : 1, scale, scale ? scale_div : 1, bias, seq, out,
                       acc, beta, B, L, N);
    return launch_status("cti_linear_residual_pb");
}

// The raw fp32 partial slabs of x (M, K) @ W^T (W (N, K) as resident planes): partials[s][m][n], s < cti_gemm_pb_partials_count(M, N, K) -- no reduce pass, no
// scale / bias: the consumer adds the slabs up as it loads them (cti_bi_pool_shift_multi_fwd).  Round 5: the unrolled BAN glimpse loop.
int cti_gemm_pb_partials_count(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    // Its own planner (the slabs are consumed raw, so a K split costs the consumer one more addend, not a reduce launch): the largest power of two with about one
    // 128 x 128 workgroup per CU and K ranges of at least 128.  plan_ksplit() leaves products of more than 96 tiles unsplit: 256 x 7 168 x 1 024, the unrolled
    // BAN loop's first product, then ran on 112 workgroups for 27 us.
    const int Kp = planes_kp(K);
    // round 6: batch-sized products run as ONE launch that splits K over a workgroup's waves (cti_gemm_skinny.hip) and leaves ONE slab
    if (gemm_skinny_enabled() && tuning_gemm_cfg() < 0 && M <= 512 && N >= 16 && Kp >= 256) return 1;
    const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
    int best = 1;
    for (int s = 2; s <= 16; s *= 2) {
        if (Kp % (s * 32) != 0 || Kp / s < 128 || tiles * s > 256) break;
        best = s;
    }
    return best;
}
size_t cti_gemm_pb_partials_workspace_bytes(int M, int K, int prec) {
    if (M <= 0 || K <= 0 || (prec != CTI_PREC_BF16X3 && prec != CTI_PREC_BF16)) return 0;
    return (planes_bytes((int64_t)M + PLANE_SLACK_ROWS, K) + 255) & ~(size_t)255;
}
int cti_gemm_pb_partials(const float* x, int64_t ldx, const void* W_planes, int M, int N, int K, int prec, float* partials, size_t partials_bytes, void* workspace,
                         size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(W_planes); CTI_REQUIRE_PTR(partials); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(M > 0 && N > 0 && K > 0 && ldx >= K, CTI_E_SHAPE, "cti_gemm_pb_partials: M=%d N=%d K=%d ldx=%lld", M, N, K, (long long)ldx);
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_gemm_pb_partials: prec=%d (resident planes exist in the bf16 modes only)", prec);
    const int S = cti_gemm_pb_partials_count(M, N, K);
    CTI_REQUIRE(partials_bytes >= sizeof(float) * (size_t)S * (size_t)M * (size_t)N, CTI_E_WORKSPACE, "cti_gemm_pb_partials: %zu bytes for %d slabs of %d x %d", partials_bytes, S, M, N);
    CTI_REQUIRE(workspace_bytes >= cti_gemm_pb_partials_workspace_bytes(M, K, prec), CTI_E_WORKSPACE, "cti_gemm_pb_partials: workspace too small");
    CTI_REQUIRE((reinterpret_cast<uintptr_t>(partials) & 15) == 0, CTI_E_ALIGN, "cti_gemm_pb_partials: the slabs must be 16-B aligned");
    hipStream_t st = as_stream(stream);
    const int Kp = planes_kp(K);
    const int64_t ra = (int64_t)M + PLANE_SLACK_ROWS, rb = (int64_t)N + PLANE_SLACK_ROWS;
    unsigned short* ah = static_cast<unsigned short*>(workspace);
    unsigned short* al_ = ah + (size_t)ra * Kp;
    const unsigned short* bh = static_cast<const unsigned short*>(W_planes);
    const unsigned short* bl = bh + (size_t)rb * Kp;
    const bool af32 = M <= 256 && (K & 3) == 0 && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;      // batch-sized: the product reads its fp32 rows itself
    int rc = CTI_OK;
    if (!af32) { rc = split_planes(x, ldx, M, K, ah, prec == CTI_PREC_BF16 ? nullptr : al_, ra, st); if (rc) return rc; }
    PlaneGemmArgs g{};
    if (af32) { g.Af = x; g.ldaf = ldx; g.Kreal = K; }
    g.Ah = ah; g.Al = al_; g.Bh = bh; g.Bl = bl; g.rows_allocA = ra; g.rows_allocB = rb; g.nb1 = 1; g.nb2 = 1;
    g.M = M; g.N = N; g.Kp = Kp; g.terms = prec == CTI_PREC_BF16X3 ? 3 : 1; g.epi = 0; g.gdiv = 1; g.scale_div = 1;
    g.C = partials; g.ldc_m = N; g.ldc_n = 1;
    if (S > 1) { g.ksplit = S; g.partial = partials; g.partials_only = 1; }       // (the plane GEMM takes the split it is given)
    else if (af32 && K != Kp) { g.Af = nullptr; rc = split_planes(x, ldx, M, K, ah, prec == CTI_PREC_BF16 ? nullptr : al_, ra, st); if (rc) return rc; }   // (a K tail: through planes)
    if (S == 1 && tuning_gemm_cfg() < 0 && gemm_skinny_eligible(g)) return gemm_skinny(g, st);
    return gemm_nt_planes(g, st);
}

int cti_seq_bcast_add(const float* x, const float* y, float* out, int B, int L, int H, void* stream) {
    CTI_REQUIRE_PTR(y); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B >= 0 && L >= 0 && H > 0, CTI_E_SHAPE, "cti_seq_bcast_add: B=%d L=%d H=%d", B, L, H);
    if ((int64_t)B * L == 0) return CTI_OK;
    hipLaunchKernelGGL(seq_bcast_add_kernel, dim3(blocks_for((int64_t)B * L * H, 256)), dim3(256), 0, as_stream(stream), x, y, out, B, L, H);
    return launch_status("cti_seq_bcast_add");
}

int cti_bce_logits_rows_fwd(const float* x, const float* target, float* row_loss, int rows, int n, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(target); CTI_REQUIRE_PTR(row_loss);
    CTI_REQUIRE(rows >= 0 && n > 0, CTI_E_SHAPE, "cti_bce_logits_rows_fwd: rows=%d n=%d", rows, n);
    if (rows == 0) return CTI_OK;
    hipLaunchKernelGGL(bce_rows_fwd_kernel, dim3(rows), dim3(256), 0, as_stream(stream), x, target, row_loss, n);
    return launch_status("cti_bce_logits_rows_fwd");
}
int cti_bce_logits_bwd(const float* x, const float* target, const float* upstream, float coef, float* dx, int64_t n, float beta,
                       void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(target); CTI_REQUIRE_PTR(dx);
    if (n <= 0) return CTI_OK;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, target, upstream, coef, dx, n, beta);
    return launch_status("cti_bce_logits_bwd");
}
int cti_kd_rows_fwd(const float* x, const float* knowledge, float* row_kl, int rows, int n, float T, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(knowledge); CTI_REQUIRE_PTR(row_kl);
    CTI_REQUIRE(rows >= 0 && n > 0 && T > 0.f, CTI_E_SHAPE, "cti_kd_rows_fwd: rows=%d n=%d T=%g", rows, n, (double)T);
    if (rows == 0) return CTI_OK;
    hipLaunchKernelGGL(kd_rows_fwd_kernel, dim3(rows), dim3(256), 0, as_stream(stream), x, knowledge, row_kl, n, 1.f / T);
    return launch_status("cti_kd_rows_fwd");
}
int cti_kd_rows_bwd(const float* x, const float* knowledge, const float* upstream, float coef, float* dx, int rows, int n, float T,
                    float beta, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(knowledge); CTI_REQUIRE_PTR(dx);
    CTI_REQUIRE(rows >= 0 && n > 0 && T > 0.f, CTI_E_SHAPE, "cti_kd_rows_bwd: rows=%d n=%d T=%g", rows, n, (double)T);
    if (rows == 0) return CTI_OK;
    hipLaunchKernelGGL(kd_rows_bwd_kernel, dim3(rows), dim3(256), 0, as_stream(stream), x, knowledge, upstream, coef, dx, n, 1.f / T, beta);
    return launch_status("cti_kd_rows_bwd");
}

}  // extern "C"
