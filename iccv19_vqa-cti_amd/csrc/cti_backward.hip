// cti_backward.hip -- primitives of the backward pass: fp32 transpose, batch sum, ReLU / bias backward, weight-norm
// gradient, and the generic strided batched NT GEMM entry the gradient contractions are expressed with.
// All byte movers: coalesced along the contiguous axis, LDS tile transpose, wave-shuffle + LDS-tree reductions.
#include "cti_common.h"
#include <cstdlib>

namespace cti {
namespace {

// ---- dst[b][c][r] = src[b][r][c]  (32 x 32 tiles through LDS, +1 padding: conflict-free both ways) -----------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int64_t ld_src, int64_t s_src,
                                                        float* __restrict__ dst, int64_t ld_dst, int64_t s_dst, int rows, int cols, int batch) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int z = blockIdx.z; z < batch; z += gridDim.z) {
        const float* s = src + (int64_t)z * s_src;
        float* d = dst + (int64_t)z * s_dst;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty + 8 * i, c = c0 + tx;
            if (r < rows && c < cols) tile[ty + 8 * i][tx] = s[(int64_t)r * ld_src + c];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, r = r0 + tx;
            if (r < rows && c < cols) d[(int64_t)c * ld_dst + r] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}

// ---- dst[i] = alpha * sum_b src[b][i] (+ beta * dst[i]) ----------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_batches_kernel(const float* __restrict__ src, float* __restrict__ dst, int nb, int64_t n,
                                                          float alpha, float beta) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    int b = 0;
    for (; b + 4 <= nb; b += 4) {                                   // four loads in flight; still one fixed summation order
        const float a0 = src[(int64_t)b * n + i], a1 = src[(int64_t)(b + 1) * n + i], a2 = src[(int64_t)(b + 2) * n + i], a3 = src[(int64_t)(b + 3) * n + i];
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; b < nb; ++b) s += src[(int64_t)b * n + i];
    dst[i] = alpha * s + (beta != 0.f ? beta * dst[i] : 0.f);
}
// many batches, few columns (the per-sample partials of a weight gradient): 64 columns x 4 batch lanes per workgroup, lanes meet in LDS
__global__ __launch_bounds__(256) void sum_batches_tall_kernel(const float* __restrict__ src, float* __restrict__ dst, int nb, int64_t n,
                                                               float alpha, float beta) {
    __shared__ float sb[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + cl;
    float s = 0.f;
    if (i < n) {
        int b = rl;
        for (; b + 12 < nb; b += 16) {
            const float a0 = src[(int64_t)b * n + i], a1 = src[(int64_t)(b + 4) * n + i], a2 = src[(int64_t)(b + 8) * n + i], a3 = src[(int64_t)(b + 12) * n + i];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; b < nb; b += 4) s += src[(int64_t)b * n + i];
    }
    sb[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && i < n) {
        const float t = (sb[0][cl] + sb[1][cl]) + (sb[2][cl] + sb[3][cl]);
        dst[i] = alpha * t + (beta != 0.f ? beta * dst[i] : 0.f);
    }
}

// ---- column sums of a tall matrix: partial[g][c] = sum of rows [g*rpg, (g+1)*rpg) of column c; 64 columns x 4 row lanes per workgroup
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ src, float* __restrict__ part, int64_t rows, int n,
                                                              int64_t rpg) {
    __shared__ float sb[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int64_t r_lo = (int64_t)blockIdx.y * rpg, r_hi = min(rows, r_lo + rpg);
    float a = 0.f;
    if (col < n)
        for (int64_t r = r_lo + rl; r < r_hi; r += 4) a += src[r * n + col];
    sb[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && col < n) part[(int64_t)blockIdx.y * n + col] = sb[0][cl] + sb[1][cl] + sb[2][cl] + sb[3][cl];
}

// ---- dzs = scale[col/div] * dy * (y > 0) [relu] ; partial column sums of the UNSCALED dz for the bias gradient ---------
// A workgroup covers CW columns x a chunk of rows with 256/CW row lanes (CW = 16, 32 or 64, picked from n so that narrow
// matrices -- the 16-column rank nets -- still use every lane); grid (ceil(n/CW), row_chunks).
template <int CW>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ scale, int scale_div, float* __restrict__ dzs,
                                                      float* __restrict__ part_b, int64_t rows, int n, int relu, int64_t rows_per_chunk) {
    constexpr int RL = 256 / CW;
    __shared__ float sb[RL][CW];
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int col = blockIdx.x * CW + cl;
    const int64_t r_lo = (int64_t)blockIdx.y * rows_per_chunk, r_hi = min(rows, r_lo + rows_per_chunk);
    float ab = 0.f;
    if (col < n) {
        const float s = scale ? scale[col / scale_div] : 1.f;
        for (int64_t r = r_lo + rl; r < r_hi; r += RL) {
            float g = dy[r * n + col];
            if (relu && !(y[r * n + col] > 0.f)) g = 0.f;
            dzs[r * n + col] = g * s;
            ab += g;
        }
    }
    sb[rl][cl] = ab;
    __syncthreads();
    if (rl == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < RL; ++i) t += sb[i][cl];
        part_b[(int64_t)blockIdx.y * n + col] = t;
    }
}

// wide matrices (n % 4 == 0, 16-B aligned): a thread owns 4 consecutive columns; 64 column groups x 4 row lanes per workgroup
__global__ __launch_bounds__(256) void act_bwd_v4_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ scale,
                                                         int scale_div, float* __restrict__ dzs, float* __restrict__ part_b, int64_t rows, int n,
                                                         int relu, int64_t rows_per_chunk) {
    __shared__ float4 sb[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cl) * 4;
    const int64_t r_lo = (int64_t)blockIdx.y * rows_per_chunk, r_hi = min(rows, r_lo + rows_per_chunk);
    float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n) {
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f);
        if (scale) s = make_float4(scale[col / scale_div], scale[(col + 1) / scale_div], scale[(col + 2) / scale_div], scale[(col + 3) / scale_div]);
        for (int64_t r = r_lo + rl; r < r_hi; r += 4) {
            float4 g = *reinterpret_cast<const float4*>(dy + r * n + col);
            if (relu) {
                const float4 yy = *reinterpret_cast<const float4*>(y + r * n + col);
                if (!(yy.x > 0.f)) g.x = 0.f;
                if (!(yy.y > 0.f)) g.y = 0.f;
                if (!(yy.z > 0.f)) g.z = 0.f;
                if (!(yy.w > 0.f)) g.w = 0.f;
            }
            *reinterpret_cast<float4*>(dzs + r * n + col) = make_float4(g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w);
            ab.x += g.x; ab.y += g.y; ab.z += g.z; ab.w += g.w;
        }
    }
    sb[rl][cl] = ab;
    __syncthreads();
    if (rl == 0 && col < n) {
        const float4 a = sb[0][cl], b = sb[1][cl], c = sb[2][cl], d = sb[3][cl];
        *reinterpret_cast<float4*>(part_b + (int64_t)blockIdx.y * n + col) =
            make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    }
}

// ---- weight-norm gradient of n_mats matrices stored back to back --------------------------------------------------------
// W = s * V, s = g / ||V||, U = x V^T, pre-activation = s U + b.  Given G = (s dz)^T x (the gradient w.r.t. V through the
// direct path, i.e. s times the gradient w.r.t. W):  dot = <G, V>;  dg = dot / g;  dV = G - (dot / ||V||^2) * V.
// One workgroup per matrix: pass 1 dot and ||V||^2 (fixed tree), pass 2 writes dV.
__global__ __launch_bounds__(1024) void wn_bwd_kernel(const float* __restrict__ dW, const float* __restrict__ V, const float* __restrict__ g,
                                                      float* __restrict__ dV, float* __restrict__ dg, int64_t elems) {
    __shared__ float p0[16], p1[16];
    const int i = blockIdx.x, t = threadIdx.x;
    const float* w = dW + (int64_t)i * elems;
    const float* v = V + (int64_t)i * elems;
    float dot = 0.f, nn = 0.f;
    for (int64_t j = t; j < elems; j += 1024) { dot = fmaf(w[j], v[j], dot); nn = fmaf(v[j], v[j], nn); }
    dot = wave_sum(dot); nn = wave_sum(nn);
    if ((t & 63) == 0) { p0[t >> 6] = dot; p1[t >> 6] = nn; }
    __syncthreads();
    float d = 0.f, n2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { d += p0[k]; n2 += p1[k]; }
    if (t == 0) dg[i] = d / g[i];
    const float c = d / n2;
    float* o = dV + (int64_t)i * elems;
    for (int64_t j = t; j < elems; j += 1024) o[j] = fmaf(-c, v[j], w[j]);
}

// ---- dropout (nn.Dropout of src/fc.py:20-21,25-26 and src/bc.py:29): Philox-4x32-10 counter RNG, 4 elements per call -------
__device__ __forceinline__ uint2 mulhilo(unsigned a, unsigned b) { const unsigned long long p = (unsigned long long)a * b; return make_uint2((unsigned)(p >> 32), (unsigned)p); }
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint2 m0 = mulhilo(0xD2511F53u, ctr.x), m1 = mulhilo(0xCD9E8D57u, ctr.z);
        ctr = make_uint4(m1.x ^ ctr.y ^ key.x, m1.y, m0.x ^ ctr.w ^ key.y, m0.y);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}
// y = x * keep / (1 - p), keep ~ Bernoulli(1 - p); mask byte stored for the backward.  backward: same kernel with x = dy, use_mask = 1.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask,
                                                      int64_t n, float p, unsigned long long seed, unsigned long long offset, int use_mask,
                                                      int64_t period, int vec, const unsigned long long* __restrict__ rng_dev) {
    // device-resident step counter: a replayed hipGraph draws fresh masks.  Its multiplier differs from the one the host applies to the SEED
    // ((seed0 * PHI + call) on the host): with the same constant, step s of a run seeded S drew the masks of step 0 of a run seeded S + s.
    if (rng_dev) seed += rng_dev[0] * 0xD1B54A32D192ED03ull;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;          // group of 4 elements
    const int64_t i0 = q * 4;
    if (i0 >= n) return;
    const float inv = 1.f / (1.f - p);
    uint4 rnd = make_uint4(0, 0, 0, 0);
    if (!use_mask) {
        const unsigned long long c = (unsigned long long)q + offset;
        rnd = philox4x32_10(make_uint4((unsigned)c, (unsigned)(c >> 32), 0u, 0u), make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    }
    const unsigned rr[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
    const unsigned thr = (unsigned)fminf(4294967295.f, p * 4294967296.f);
    // fast path: the whole group of 4 is in range and 16-B / 4-B aligned in x, y and mask (period a multiple of 4: the group does not
    // straddle two copies): one 16-B load, one 16-B store, one 4-B mask access, one 64-bit modulo per group instead of per element
    const int64_t s0 = period ? i0 % period : i0;
    if (!y) {                                                           // mask only (cti_ranknets_drop_* apply it where they form their operands)
        if (i0 + 4 <= n && (reinterpret_cast<uintptr_t>(mask) & 3) == 0) {
            *reinterpret_cast<uchar4*>(mask + i0) = make_uchar4(rr[0] >= thr, rr[1] >= thr, rr[2] >= thr, rr[3] >= thr);
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) mask[i0 + j] = rr[j] >= thr ? 1 : 0;
        return;
    }
    if (vec && i0 + 4 <= n) {
        const float4 xv = *reinterpret_cast<const float4*>(x + s0);
        uchar4 mk;
        if (use_mask) mk = *reinterpret_cast<const uchar4*>(mask + i0);
        else {
            mk = make_uchar4(rr[0] >= thr, rr[1] >= thr, rr[2] >= thr, rr[3] >= thr);
            *reinterpret_cast<uchar4*>(mask + i0) = mk;
        }
        *reinterpret_cast<float4*>(y + i0) = make_float4(mk.x ? xv.x * inv : 0.f, mk.y ? xv.y * inv : 0.f, mk.z ? xv.z * inv : 0.f, mk.w ? xv.w * inv : 0.f);
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + j;
        if (i < n) {
            const bool keep = use_mask ? mask[i] != 0 : rr[j] >= thr;
            if (!use_mask) mask[i] = keep ? 1 : 0;
            y[i] = keep ? x[period ? i % period : i] * inv : 0.f;
        }
    }
}

// mask only, 16 bytes per thread (four Philox calls, the SAME counters as dropout_kernel: counter = element / 4) and one 16-B store: the 4-B
// stores of the general kernel wrote the 151 MB mask of the visual rank nets at 1.9 TB/s
__global__ __launch_bounds__(256) void dropout_mask16_kernel(uint8_t* __restrict__ mask, int64_t n16, float p, unsigned long long seed,
                                                             unsigned long long offset, const unsigned long long* __restrict__ rng_dev) {
    if (rng_dev) seed += rng_dev[0] * 0xD1B54A32D192ED03ull;               // (see dropout_kernel)
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;          // group of 16 elements
    if (t >= n16) return;
    const unsigned thr = (unsigned)fminf(4294967295.f, p * 4294967296.f);
    unsigned w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned long long c = (unsigned long long)(t * 4 + u) + offset;
        const uint4 r = philox4x32_10(make_uint4((unsigned)c, (unsigned)(c >> 32), 0u, 0u), make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
        w[u] = (r.x >= thr ? 1u : 0u) | (r.y >= thr ? 0x100u : 0u) | (r.z >= thr ? 0x10000u : 0u) | (r.w >= thr ? 0x1000000u : 0u);
    }
    *reinterpret_cast<uint4*>(mask + t * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}

// multi-workgroup form for large matrices: partial <G,V> and ||V||^2 per 4,096-element chunk, then every workgroup of the
// apply pass re-reduces its matrix's partials in the same fixed order
constexpr int64_t WNB_CHUNK = 4096;
__global__ __launch_bounds__(256) void wn_bwd_partial_kernel(const float* __restrict__ G, const float* __restrict__ V, float* __restrict__ part,
                                                             int64_t elems, int cpm) {
    __shared__ float r0[4], r1[4];
    const int mat = blockIdx.y, ch = blockIdx.x;
    const float* w = G + (int64_t)mat * elems; const float* v = V + (int64_t)mat * elems;
    const int64_t lo = (int64_t)ch * WNB_CHUNK, hi = min(elems, lo + WNB_CHUNK);
    float d = 0.f, n = 0.f;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) { d = fmaf(w[j], v[j], d); n = fmaf(v[j], v[j], n); }
    d = wave_sum(d); n = wave_sum(n);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = d; r1[threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((int64_t)mat * cpm + ch) * 2] = (r0[0] + r0[1]) + (r0[2] + r0[3]);
        part[((int64_t)mat * cpm + ch) * 2 + 1] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    }
}
__global__ __launch_bounds__(256) void wn_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ V, const float* __restrict__ g,
                                                           const float* __restrict__ part, float* __restrict__ dV, float* __restrict__ dg,
                                                           int64_t elems, int cpm, int vec) {
    __shared__ float r0[4], r1[4];
    const int mat = blockIdx.y, ch = blockIdx.x;
    float d = 0.f, n2 = 0.f;                                        // every workgroup re-reduces its matrix's partials in the same fixed tree
    for (int c = threadIdx.x; c < cpm; c += 256) { d += part[((int64_t)mat * cpm + c) * 2]; n2 += part[((int64_t)mat * cpm + c) * 2 + 1]; }
    d = wave_sum(d); n2 = wave_sum(n2);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = d; r1[threadIdx.x >> 6] = n2; }
    __syncthreads();
    d = (r0[0] + r0[1]) + (r0[2] + r0[3]); n2 = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    if (ch == 0 && threadIdx.x == 0) dg[mat] = d / g[mat];
    const float c = d / n2;
    const int64_t lo = (int64_t)ch * WNB_CHUNK, hi = min(elems, lo + WNB_CHUNK);
    const float* w = G + (int64_t)mat * elems; const float* v = V + (int64_t)mat * elems;
    float* o = dV + (int64_t)mat * elems;
    if (vec) {                                                      // elems % 4 == 0 and 16-B aligned bases: whole float4s only
        for (int64_t j = lo + 4 * (int64_t)threadIdx.x; j < hi; j += 1024) {
            const float4 ww = *reinterpret_cast<const float4*>(w + j), vv = *reinterpret_cast<const float4*>(v + j);
            *reinterpret_cast<float4*>(o + j) = make_float4(fmaf(-c, vv.x, ww.x), fmaf(-c, vv.y, ww.y), fmaf(-c, vv.z, ww.z), fmaf(-c, vv.w, ww.w));
        }
        return;
    }
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) o[j] = fmaf(-c, v[j], w[j]);
}

}  // namespace
}  // namespace cti

using namespace cti;

__global__ void counter_add_kernel(long long* ctr, long long inc) { if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += inc; }

extern "C" int cti_counter_add(int64_t* counter, int64_t inc, void* stream) {
    CTI_REQUIRE_PTR(counter);
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, as_stream(stream), reinterpret_cast<long long*>(counter), (long long)inc);
    return launch_status("cti_counter_add");
}

extern "C" int cti_dropout_g(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, int use_mask,
                             int64_t period, const uint64_t* rng_dev, void* stream);
extern "C" int cti_dropout(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, int use_mask,
                           int64_t period, void* stream) {
    return cti_dropout_g(x, y, mask, n, p, seed, offset, use_mask, period, nullptr, stream);
}

extern "C" int cti_dropout_g(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, int use_mask,
                             int64_t period, const uint64_t* rng_dev, void* stream) {
    CTI_REQUIRE_PTR(mask);
    if (y || use_mask) { CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(y); }
    CTI_REQUIRE(n > 0 && p >= 0.f && p < 1.f && period >= 0, CTI_E_SHAPE, "cti_dropout: n=%lld p=%f", (long long)n, p);
    if (!y && !use_mask && n >= 16 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0) {          // mask only: 16-B stores for the bulk, the tail below
        const int64_t n16 = n / 16;
        hipLaunchKernelGGL(dropout_mask16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, as_stream(stream), mask, n16, p,
                           (unsigned long long)seed, (unsigned long long)offset, reinterpret_cast<const unsigned long long*>(rng_dev));
        int rc = launch_status("cti_dropout/mask16"); if (rc) return rc;
        if (n16 * 16 == n) return 0;
        mask += n16 * 16; offset += (uint64_t)n16 * 4; n -= n16 * 16;
    }
    const int64_t groups = (n + 3) / 4;
    const int vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && (reinterpret_cast<uintptr_t>(mask) & 3) == 0 &&
                    (period & 3) == 0;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, as_stream(stream), x, y, mask, n, p,
                       (unsigned long long)seed, (unsigned long long)offset, use_mask, period, vec, reinterpret_cast<const unsigned long long*>(rng_dev));
    return launch_status("cti_dropout");
}

extern "C" int cti_transpose_f32(const float* src, int64_t ld_src, int64_t batch_stride_src, float* dst, int64_t ld_dst,
                                 int64_t batch_stride_dst, int rows, int cols, int batch, void* stream) {
    CTI_REQUIRE_PTR(src); CTI_REQUIRE_PTR(dst);
    CTI_REQUIRE(rows > 0 && cols > 0 && batch > 0 && ld_src >= cols && ld_dst >= rows, CTI_E_SHAPE,
                "cti_transpose_f32: rows=%d cols=%d batch=%d ld_src=%lld ld_dst=%lld", rows, cols, batch, (long long)ld_src, (long long)ld_dst);
    CTI_REQUIRE((rows + 31) / 32 <= 65535, CTI_E_SHAPE, "cti_transpose_f32: rows=%d too large for grid.y", rows);
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch < 32768 ? batch : 32768);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, as_stream(stream), src, ld_src, batch_stride_src, dst, ld_dst,
                       batch_stride_dst, rows, cols, batch);
    return launch_status("cti_transpose_f32");
}

extern "C" int cti_sum_batches(const float* src, float* dst, int nb, int64_t n, float alpha, float beta, void* stream) {
    CTI_REQUIRE_PTR(src); CTI_REQUIRE_PTR(dst);
    CTI_REQUIRE(nb > 0 && n > 0, CTI_E_SHAPE, "cti_sum_batches: nb=%d n=%lld", nb, (long long)n);
    if (nb >= 32 && n < (1 << 20))
        hipLaunchKernelGGL(sum_batches_tall_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, as_stream(stream), src, dst, nb, n, alpha, beta);
    else
        hipLaunchKernelGGL(sum_batches_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), src, dst, nb, n, alpha, beta);
    return launch_status("cti_sum_batches");
}

static int col_groups(int64_t rows) { int64_t g = (rows + 31) / 32; return (int)(g < 1 ? 1 : (g > 256 ? 256 : g)); }

extern "C" size_t cti_col_sum_workspace_bytes(int64_t rows, int n) {
    if (rows <= 0 || n <= 0) return 0;
    return sizeof(float) * (size_t)col_groups(rows) * n;
}

extern "C" int cti_col_sum(const float* src, int64_t rows, int n, float* dst, float alpha, float beta, void* workspace, size_t workspace_bytes,
                           void* stream) {
    CTI_REQUIRE_PTR(src); CTI_REQUIRE_PTR(dst); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(rows > 0 && n > 0, CTI_E_SHAPE, "cti_col_sum: rows=%lld n=%d", (long long)rows, n);
    CTI_REQUIRE(workspace_bytes >= cti_col_sum_workspace_bytes(rows, n), CTI_E_WORKSPACE, "cti_col_sum: workspace too small");
    const int groups = col_groups(rows);
    const int64_t rpg = (rows + groups - 1) / groups;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3((n + 63) / 64, groups), dim3(256), 0, as_stream(stream), src, part, rows, n, rpg);
    int rc = launch_status("cti_col_sum"); if (rc) return rc;
    return cti_sum_batches(part, dst, groups, n, alpha, beta, stream);
}

static int act_chunks(int64_t rows) { int64_t c = (rows + 63) / 64; return (int)(c < 1 ? 1 : (c > 1024 ? 1024 : c)); }   // x column blocks = workgroups

extern "C" size_t cti_act_bwd_workspace_bytes(int64_t rows, int n) {
    if (rows <= 0 || n <= 0) return 0;
    return sizeof(float) * (size_t)act_chunks(rows) * n;
}

extern "C" int cti_act_bwd(const float* dy, const float* y, const float* scale, int scale_div, float* dzs, float* dbias, int64_t rows,
                           int n, int act, void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(dy); CTI_REQUIRE_PTR(y); CTI_REQUIRE_PTR(dzs); CTI_REQUIRE_PTR(dbias); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(rows > 0 && n > 0 && (scale == nullptr || scale_div > 0), CTI_E_SHAPE, "cti_act_bwd: rows=%lld n=%d", (long long)rows, n);
    CTI_REQUIRE(workspace_bytes >= cti_act_bwd_workspace_bytes(rows, n), CTI_E_WORKSPACE, "cti_act_bwd: workspace too small");
    const int chunks = act_chunks(rows);
    const int64_t rpc = (rows + chunks - 1) / chunks;
    float* pb = static_cast<float*>(workspace);
    const int relu = act == CTI_ACT_RELU ? 1 : 0, sdiv = scale ? scale_div : 1;
    const bool v4 = n >= 256 && n % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dzs) |
                                                 reinterpret_cast<uintptr_t>(pb)) & 15) == 0;
    if (v4)           hipLaunchKernelGGL(act_bwd_v4_kernel, dim3((n / 4 + 63) / 64, chunks), dim3(256), 0, as_stream(stream), dy, y, scale, sdiv, dzs, pb, rows, n, relu, rpc);
    else if (n <= 16) hipLaunchKernelGGL(act_bwd_kernel<16>, dim3((n + 15) / 16, chunks), dim3(256), 0, as_stream(stream), dy, y, scale, sdiv, dzs, pb, rows, n, relu, rpc);
    else if (n <= 32) hipLaunchKernelGGL(act_bwd_kernel<32>, dim3((n + 31) / 32, chunks), dim3(256), 0, as_stream(stream), dy, y, scale, sdiv, dzs, pb, rows, n, relu, rpc);
    else              hipLaunchKernelGGL(act_bwd_kernel<64>, dim3((n + 63) / 64, chunks), dim3(256), 0, as_stream(stream), dy, y, scale, sdiv, dzs, pb, rows, n, relu, rpc);
    int rc = launch_status("cti_act_bwd"); if (rc) return rc;
    return cti_sum_batches(pb, dbias, chunks, n, 1.f, 0.f, stream);
}

extern "C" size_t cti_wn_bwd_workspace_bytes(int n_mats, int64_t elems) {
    if (n_mats <= 0 || elems <= WNB_CHUNK) return 0;
    return sizeof(float) * 2 * (size_t)n_mats * (size_t)((elems + WNB_CHUNK - 1) / WNB_CHUNK);
}

extern "C" int cti_wn_bwd(const float* G, const float* weight_v, const float* weight_g, float* dweight_v, float* dweight_g,
                          int n_mats, int64_t elems, void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(G); CTI_REQUIRE_PTR(weight_v); CTI_REQUIRE_PTR(weight_g); CTI_REQUIRE_PTR(dweight_v); CTI_REQUIRE_PTR(dweight_g);
    CTI_REQUIRE(n_mats > 0 && n_mats <= 65535 && elems > 0, CTI_E_SHAPE, "cti_wn_bwd: n_mats=%d elems=%lld", n_mats, (long long)elems);
    if (elems <= WNB_CHUNK || workspace == nullptr) {
        hipLaunchKernelGGL(wn_bwd_kernel, dim3(n_mats), dim3(1024), 0, as_stream(stream), G, weight_v, weight_g, dweight_v, dweight_g, elems);
        return launch_status("cti_wn_bwd");
    }
    CTI_REQUIRE(workspace_bytes >= cti_wn_bwd_workspace_bytes(n_mats, elems), CTI_E_WORKSPACE, "cti_wn_bwd: workspace too small");
    const int cpm = (int)((elems + WNB_CHUNK - 1) / WNB_CHUNK);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(wn_bwd_partial_kernel, dim3(cpm, n_mats), dim3(256), 0, as_stream(stream), G, weight_v, part, elems, cpm);
    int rc = launch_status("cti_wn_bwd/partial"); if (rc) return rc;
    const int vec = elems % 4 == 0 && ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(weight_v) | reinterpret_cast<uintptr_t>(dweight_v)) & 15) == 0;
    hipLaunchKernelGGL(wn_bwd_apply_kernel, dim3(cpm, n_mats), dim3(256), 0, as_stream(stream), G, weight_v, weight_g, part, dweight_v, dweight_g,
                       elems, cpm, vec);
    return launch_status("cti_wn_bwd/apply");
}

// ---- generic strided batched NT GEMM: C[z][m,n] = act(scale[n/div] * sum_k A[z][m,k] * B[z][n,k] + bias[n]) ---------------
// C (N x K) = a^T b for a (M x N), b (M x K) row-major: the weight-gradient contraction over the ROW axis.  Both operands go
// straight to transposed bf16 planes (split_planes_t), the M axis is cut into S ranges that run as extra workgroups, a reduce kernel
// sums the S partials.  (The previous route wrote transposed fp32 copies of both operands first.)
static void tn_plan(int64_t M, int N, int K, int* S, int64_t* Mp) {
    *S = plan_ksplit_tn(M, N, K);
    const int64_t q = 32 * (int64_t)*S;
    *Mp = (M + q - 1) / q * q;
}
extern "C" size_t cti_gemm_tn_workspace_bytes(int64_t M, int N, int K, int prec) {
    if (prec == CTI_PREC_F32 || M <= 0 || N <= 0 || K <= 0) return 0;
    int S; int64_t Mp; tn_plan(M, N, K, &S, &Mp);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t pa = 2 * sizeof(unsigned short) * (size_t)(N + PLANE_SLACK_ROWS) * Mp, pb = 2 * sizeof(unsigned short) * (size_t)(K + PLANE_SLACK_ROWS) * Mp;
    return al(pa) + al(pb) + (S > 1 ? al(sizeof(float) * (size_t)S * N * K) : 0);
}
extern "C" int cti_gemm_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* C, int64_t M, int N, int K, int prec, void* workspace,
                           size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(a); CTI_REQUIRE_PTR(b); CTI_REQUIRE_PTR(C); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= N && ldb >= K, CTI_E_SHAPE, "cti_gemm_tn: M=%lld N=%d K=%d", (long long)M, N, K);
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_gemm_tn: prec=%d (the exact-fp32 mode contracts transposed fp32 copies with cti_gemm_nt)", prec);
    CTI_REQUIRE(workspace_bytes >= cti_gemm_tn_workspace_bytes(M, N, K, prec), CTI_E_WORKSPACE, "cti_gemm_tn: workspace too small");
    int S; int64_t Mp; tn_plan(M, N, K, &S, &Mp);
    CTI_REQUIRE(Mp / 16 <= 65535, CTI_E_SHAPE, "cti_gemm_tn: M=%lld exceeds the split grid", (long long)M);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int64_t ra = (int64_t)N + PLANE_SLACK_ROWS, rb = (int64_t)K + PLANE_SLACK_ROWS;
    char* w = static_cast<char*>(workspace);
    unsigned short* ah = reinterpret_cast<unsigned short*>(w);            unsigned short* al_ = ah + (size_t)ra * Mp;
    w += al(2 * sizeof(unsigned short) * (size_t)ra * Mp);
    unsigned short* bh = reinterpret_cast<unsigned short*>(w);            unsigned short* bl = bh + (size_t)rb * Mp;
    w += al(2 * sizeof(unsigned short) * (size_t)rb * Mp);
    float* part = reinterpret_cast<float*>(w);
    hipStream_t st = as_stream(stream);
    int rc = split_planes_t(a, lda, M, N, Mp, ah, al_, ra, st); if (rc) return rc;
    rc = split_planes_t(b, ldb, M, K, Mp, bh, bl, rb, st); if (rc) return rc;
    PlaneGemmArgs g{};
    g.Ah = ah; g.Al = al_; g.Bh = bh; g.Bl = bl; g.rows_allocA = ra; g.rows_allocB = rb; g.nb1 = 1; g.nb2 = 1;
    g.M = N; g.N = K; g.Kp = (int)Mp; g.terms = prec == CTI_PREC_BF16X3 ? 3 : 1; g.epi = 0;
    g.C = C; g.ldc_m = K; g.ldc_n = 1; g.scale_div = 1;
    if (S > 1) { g.ksplit = S; g.partial = part; }
    return gemm_nt_planes(g, st);
}

// ---- resident operand planes: a weight matrix split once, reused by every forward until it changes -----------------------------------
// Block layout: [hi | lo], each (rows + PLANE_SLACK_ROWS) x Kp bf16, chunk-major.
extern "C" size_t cti_operand_planes_bytes(int64_t rows, int K) {
    if (rows <= 0 || K <= 0) return 0;
    return planes_bytes(rows + PLANE_SLACK_ROWS, K);
}
extern "C" int cti_split_operand(const float* x, int64_t ld, int64_t rows, int K, void* planes, size_t planes_bytes_, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(planes);
    CTI_REQUIRE(rows > 0 && K > 0 && ld >= K, CTI_E_SHAPE, "cti_split_operand: rows=%lld K=%d ld=%lld", (long long)rows, K, (long long)ld);
    CTI_REQUIRE(planes_bytes_ >= cti_operand_planes_bytes(rows, K), CTI_E_WORKSPACE, "cti_split_operand: block too small");
    const int64_t ra = rows + PLANE_SLACK_ROWS;
    unsigned short* hi = static_cast<unsigned short*>(planes);
    return split_planes(x, ld, rows, K, hi, hi + (size_t)ra * planes_kp(K), ra, as_stream(stream));
}

// C (rows x K) = a (rows x N) . b (N x K), both row-major: the input gradient dx = dzs . V of a Linear layer.  b goes straight to the planes of
// b^T through the transposing split (its contraction axis N is the ROW axis of b), so no transposed fp32 copy of the weight is written first.
extern "C" size_t cti_gemm_nn_workspace_bytes(int64_t rows, int N, int K, int prec) {
    if (prec == CTI_PREC_F32 || rows <= 0 || N <= 0 || K <= 0) return 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t n = al(planes_bytes(rows + PLANE_SLACK_ROWS, N)) + al(planes_bytes((int64_t)K + PLANE_SLACK_ROWS, N));
    if (rows < (1ll << 31)) {
        const int S = plan_ksplit((int)rows, K, planes_kp(N), 1);
        if (S > 1) n += al(sizeof(float) * (size_t)S * (size_t)rows * (size_t)K);
    }
    return n;
}
extern "C" int cti_gemm_nn(const float* a, int64_t lda, const float* b, int64_t ldb, float* C, int64_t rows, int N, int K, int prec, void* workspace,
                           size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(a); CTI_REQUIRE_PTR(b); CTI_REQUIRE_PTR(C); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(rows > 0 && rows < (1ll << 31) && N > 0 && K > 0 && lda >= N && ldb >= K, CTI_E_SHAPE, "cti_gemm_nn: rows=%lld N=%d K=%d", (long long)rows, N, K);
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_gemm_nn: prec=%d (the exact-fp32 mode multiplies a transposed fp32 copy with cti_gemm_nt)", prec);
    CTI_REQUIRE(workspace_bytes >= cti_gemm_nn_workspace_bytes(rows, N, K, prec), CTI_E_WORKSPACE, "cti_gemm_nn: workspace too small");
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int Np = planes_kp(N);
    const int64_t ra = rows + PLANE_SLACK_ROWS, rb = (int64_t)K + PLANE_SLACK_ROWS;
    char* w = static_cast<char*>(workspace);
    unsigned short* ah = reinterpret_cast<unsigned short*>(w);            unsigned short* al_ = ah + (size_t)ra * Np;
    w += al(planes_bytes(ra, N));
    unsigned short* bh = reinterpret_cast<unsigned short*>(w);            unsigned short* bl = bh + (size_t)rb * Np;
    w += al(planes_bytes(rb, N));
    hipStream_t st = as_stream(stream);
    int rc = split_planes(a, lda, rows, N, ah, al_, ra, st); if (rc) return rc;
    rc = split_planes_t(b, ldb, N, K, Np, bh, bl, rb, st); if (rc) return rc;
    PlaneGemmArgs g{};
    g.Ah = ah; g.Al = al_; g.Bh = bh; g.Bl = bl; g.rows_allocA = ra; g.rows_allocB = rb; g.nb1 = 1; g.nb2 = 1;
    g.M = (int)rows; g.N = K; g.Kp = Np; g.terms = prec == CTI_PREC_BF16X3 ? 3 : 1; g.epi = 0;
    g.C = C; g.ldc_m = K; g.ldc_n = 1; g.scale_div = 1;
    const int S = plan_ksplit((int)rows, K, Np, 1);
    if (S > 1) { g.ksplit = S; g.partial = reinterpret_cast<float*>(w); }
    return gemm_nt_planes(g, st);
}

// CTI_AF32_PB=1 (experiment; built, under test, measured SLOWER and therefore off): products against resident planes below 128 tiles of 256 x 256 read their
// fp32 A rows directly instead of a split launch in front of each -- MC CTI forward 0.910 -> 0.940 ms, FFOE BAN + CTI 2.558 -> 2.600 ms: the 41 splits cost
// 5-7 us each, but every column tile of the small-tile kernel then moves twice the A bytes through LDS-DMA and converts the same rows again
#ifndef CTI_AF32_ROWS_DEFAULT
#define CTI_AF32_ROWS_DEFAULT 256      // measured (tools/ab_af32_rows.sh): c3 296.6 -> 304 k samples/s, c4 137.8 -> 139.0 k; 512 and 4096 give less
#endif
static bool af32_pb() { static const bool v = [] { const char* e = getenv("CTI_AF32_PB"); return e && e[0] == '1'; }(); return v; }
// the fp32-A form for SMALL row counts only (the batch-sized products of the model forwards' dependent chains: their split launch costs as much as they do)
static int af32_rows() { static const int v = [] { const char* e = getenv("CTI_AF32_ROWS"); return e ? atoi(e) : CTI_AF32_ROWS_DEFAULT; }(); return v; }

constexpr size_t PB_BATCHED_PARTIALS = sizeof(float) * 288 * 128 * 128;
// cti_gemm_nt with the B operand given as resident planes (cti_split_operand of the (rowsB_total x K) matrix): only A is split here.
extern "C" size_t cti_gemm_nt_pb_workspace_bytes(int64_t rowsA_total, int64_t rowsB_total, int K, int prec) {
    if (prec == CTI_PREC_F32 || rowsA_total <= 0 || rowsB_total <= 0 || K <= 0) return 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t n = al(planes_bytes(rowsA_total + PLANE_SLACK_ROWS, K));
    if (rowsA_total < (1ll << 31) && rowsB_total < (1ll << 31)) {
        const int S = plan_ksplit((int)rowsA_total, (int)rowsB_total, planes_kp(K), 1);
        if (S > 1) n += al(sizeof(float) * (size_t)S * (size_t)rowsA_total * (size_t)rowsB_total);
    }
    // batches of skinny products (round 4: split K too): the planner never asks for more than 288 tiles of 128 x 128 partials
    if (rowsA_total <= 8192) n = n > al(planes_bytes(rowsA_total + PLANE_SLACK_ROWS, K)) + PB_BATCHED_PARTIALS ? n : al(planes_bytes(rowsA_total + PLANE_SLACK_ROWS, K)) + PB_BATCHED_PARTIALS;
    return n;
}
// The exact need of ONE call (round 5, ADVICE r4: the bound above adds 18.9 MB of batched-partials room to every product of <= 8 192 rows, also to the
// unbatched ones that never split that way, once per stream pool): A planes + the partials the call's own plan writes.
extern "C" size_t cti_gemm_nt_pb_workspace_bytes2(int64_t rowsA_total, int64_t rowsB_total, int K, int prec, int nb1, int M, int N) {
    if (prec == CTI_PREC_F32 || rowsA_total <= 0 || rowsB_total <= 0 || K <= 0 || nb1 <= 0 || M <= 0 || N <= 0) return 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t n = al(planes_bytes(rowsA_total + PLANE_SLACK_ROWS, K));
    int S = 1;
    if (nb1 == 1 && M == rowsA_total && N == rowsB_total) S = plan_ksplit(M, N, planes_kp(K), 1);
    else if (nb1 > 1 && rowsA_total <= 8192) S = plan_ksplit(M, N, planes_kp(K), nb1);
    if (S > 1) n += al(sizeof(float) * (size_t)S * (size_t)nb1 * (size_t)M * (size_t)N);
    return n;
}
extern "C" int cti_gemm_nt_pb(const float* A, int64_t lda, int64_t rowsA_total, int64_t rA1, const void* B_planes, int64_t rowsB_total, int64_t rB1,
                              float* C, int64_t ldc_m, int64_t sC1, int nb1, int M, int N, int K, const float* scale, int scale_div, int64_t scale_bs,
                              const float* bias, int64_t bias_bs, int act, int prec, void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(A); CTI_REQUIRE_PTR(B_planes); CTI_REQUIRE_PTR(C); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(M > 0 && N > 0 && K > 0 && nb1 > 0 && lda >= K, CTI_E_SHAPE, "cti_gemm_nt_pb: M=%d N=%d K=%d nb=%d", M, N, K, nb1);
    CTI_REQUIRE((int64_t)(nb1 - 1) * rA1 + M <= rowsA_total && (int64_t)(nb1 - 1) * rB1 + N <= rowsB_total, CTI_E_SHAPE, "cti_gemm_nt_pb: batches run past the operand rows");
    CTI_REQUIRE(act == CTI_ACT_NONE || act == CTI_ACT_RELU, CTI_E_UNSUPPORTED, "cti_gemm_nt_pb: act=%d", act);
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_gemm_nt_pb: prec=%d (resident planes exist in the bf16 modes only)", prec);
    CTI_REQUIRE(workspace_bytes >= cti_gemm_nt_pb_workspace_bytes2(rowsA_total, rowsB_total, K, prec, nb1, M, N), CTI_E_WORKSPACE, "cti_gemm_nt_pb: workspace %zu < %zu", workspace_bytes,
                cti_gemm_nt_pb_workspace_bytes2(rowsA_total, rowsB_total, K, prec, nb1, M, N));
    const int Kp = planes_kp(K);
    const int64_t ra = rowsA_total + PLANE_SLACK_ROWS, rb = rowsB_total + PLANE_SLACK_ROWS;
    unsigned short* ah = static_cast<unsigned short*>(workspace);
    unsigned short* al = ah + (size_t)ra * Kp;
    const unsigned short* bh = static_cast<const unsigned short*>(B_planes);
    const unsigned short* bl = bh + (size_t)rb * Kp;
    // fp32 A operand read as it stands (no split launch, no A planes) wherever the product is not one of the big ones (those take the 256 x 256 tiles:
    // plain bf16 -> cti_gemm16.hip on a hi plane; bf16x3 -> the planes kernel, whose fp32-A form measured slower at that size)
    const long long tiles256 = (long long)nb1 * ((M + 255) / 256) * ((N + 255) / 256);
    // (round 6: batches of batch-sized products too -- cti_gemm_skinny.hip reads the fp32 rows of every batch itself: no split launch in front of it)
    const bool skinny_rows = gemm_skinny_enabled() && tuning_gemm_cfg() < 0 && M <= 512 && K == Kp && Kp >= 256 && (int64_t)nb1 * M <= 4 * af32_rows() && rowsA_total <= 8192;      // (exactly the products gemm_nt_planes() hands to that kernel)
    const bool af32 = (af32_pb() || (int64_t)nb1 * M <= af32_rows() || skinny_rows) && tiles256 < 128 && (K & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
    int rc = CTI_OK;
    if (!af32) { rc = split_planes(A, lda, rowsA_total, K, ah, prec == CTI_PREC_BF16 ? nullptr : al, ra, as_stream(stream)); if (rc) return rc; }   // plain bf16: the products read the hi plane only
    PlaneGemmArgs g{};
    if (af32) { g.Af = A; g.ldaf = lda; g.Kreal = K; }
    g.Ah = ah; g.Al = al; g.Bh = bh; g.Bl = bl; g.rows_allocA = ra; g.rows_allocB = rb;
    g.rA1 = rA1; g.rB1 = rB1; g.nb1 = nb1; g.nb2 = 1;
    g.M = M; g.N = N; g.Kp = Kp; g.terms = prec == CTI_PREC_BF16X3 ? 3 : 1;
    g.C = C; g.ldc_m = ldc_m; g.ldc_n = 1; g.sC1 = sC1; g.epi = 0; g.gdiv = 1;
    g.scale = scale; g.scale_div = scale ? scale_div : 1; g.bias = bias; g.relu = act == CTI_ACT_RELU;
    g.scale_bs = scale_bs; g.bias_bs = bias_bs;
    if (nb1 == 1 && M == rowsA_total && N == rowsB_total) {
        const int S = plan_ksplit(M, N, Kp, 1);
        if (S > 1) { g.ksplit = S; g.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + ((planes_bytes(ra, K) + 255) & ~(size_t)255)); }
    } else if (nb1 > 1 && rowsA_total <= 8192) {
        // a batch of skinny products (the two sequences of the CTI models' glimpse loop: 2 x (256 x 1 024 x 1 024)): K ranges as a second batch axis, one reduce pass
        const int S = plan_ksplit(M, N, Kp, nb1);
        const size_t off = (planes_bytes(ra, K) + 255) & ~(size_t)255;
        if (S > 1 && off + sizeof(float) * (size_t)S * nb1 * M * N <= workspace_bytes) { g.ksplit = S; g.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + off); }
    }
    return gemm_nt_planes(g, as_stream(stream));
}

// Plain-bf16 product whose A operand is a row-major bf16 matrix as it stands (no split pass) against resident planes; C as fp32 rows or as
// bf16 rows -- the next layer's A operand.  cti_gemm16.hip; K % 32 == 0 (a row is read in whole 64-B stages).  With a workspace
// (cti_gemm_bf16_rows_sk_workspace_bytes(); zeroed once, one per stream) products whose tiles do not fill whole rounds of the compute units are cut stream-K.
static int gemm_bf16_rows_impl(const char* who, const void* A_bf16, int64_t lda, int64_t rowsA_total, int64_t rA1, const void* B_planes, int64_t rowsB_total, int64_t rB1,
                               void* C, int c_bf16, int64_t ldc_m, int64_t sC1, int nb1, int M, int N, int K, const float* scale, int scale_div,
                               int64_t scale_bs, const float* bias, int64_t bias_bs, int act, void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE(A_bf16 && B_planes && C, CTI_E_NULL, "%s: A_bf16 / B_planes / C is NULL", who);
    CTI_REQUIRE(M > 0 && N > 0 && K > 0 && nb1 > 0 && lda >= K, CTI_E_SHAPE, "%s: M=%d N=%d K=%d nb=%d", who, M, N, K, nb1);
    CTI_REQUIRE(K % 32 == 0, CTI_E_UNSUPPORTED, "%s: K=%d must be a multiple of 32 (rows are read in 64-B stages; zero-pad the operand)", who, K);
    CTI_REQUIRE((int64_t)(nb1 - 1) * rA1 + M <= rowsA_total && (int64_t)(nb1 - 1) * rB1 + N <= rowsB_total, CTI_E_SHAPE, "%s: batches run past the operand rows", who);
    CTI_REQUIRE(act == CTI_ACT_NONE || act == CTI_ACT_RELU, CTI_E_UNSUPPORTED, "%s: act=%d", who, act);
    CTI_REQUIRE(!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) == 0, CTI_E_ALIGN, "%s: the stream-K workspace must be 256-B aligned", who);
    const int64_t rb = rowsB_total + PLANE_SLACK_ROWS;
    PlaneGemmArgs g{};
    g.Abf = A_bf16; g.ldabf = lda; g.Bh = static_cast<const unsigned short*>(B_planes); g.Bl = g.Bh + (size_t)rb * planes_kp(K);
    g.rows_allocA = rowsA_total; g.rows_allocB = rb; g.rA1 = rA1; g.rB1 = rB1; g.nb1 = nb1; g.nb2 = 1;
    g.M = M; g.N = N; g.Kp = K; g.terms = 1; g.epi = c_bf16 ? 5 : 0; g.gdiv = 1;
    g.C = static_cast<float*>(C); g.ldc_m = ldc_m; g.ldc_n = 1; g.sC1 = sC1;
    g.scale = scale; g.scale_div = scale ? scale_div : 1; g.bias = bias; g.relu = act == CTI_ACT_RELU; g.scale_bs = scale_bs; g.bias_bs = bias_bs;
    g.sk_ws = workspace; g.sk_ws_bytes = workspace ? workspace_bytes : 0;
    CTI_REQUIRE(gemm16_eligible(g), CTI_E_ALIGN, "%s: A needs 16-B aligned rows (lda %% 8 == 0), C rows of a multiple of 4 elements at a %d-B aligned origin, K / 32 >= 4 with bias or scale",
                who, c_bf16 ? 8 : 16);
    return gemm16_planes(g, as_stream(stream));
}

extern "C" int cti_gemm_bf16_rows(const void* A_bf16, int64_t lda, int64_t rowsA_total, int64_t rA1, const void* B_planes, int64_t rowsB_total, int64_t rB1,
                                  void* C, int c_bf16, int64_t ldc_m, int64_t sC1, int nb1, int M, int N, int K, const float* scale, int scale_div,
                                  int64_t scale_bs, const float* bias, int64_t bias_bs, int act, void* stream) {
    return gemm_bf16_rows_impl("cti_gemm_bf16_rows", A_bf16, lda, rowsA_total, rA1, B_planes, rowsB_total, rB1, C, c_bf16, ldc_m, sC1, nb1, M, N, K, scale, scale_div, scale_bs,
                               bias, bias_bs, act, nullptr, 0, stream);
}

extern "C" size_t cti_gemm_bf16_rows_sk_workspace_bytes(void) { return gemm16_sk_workspace_bytes(); }

extern "C" int cti_gemm_bf16_rows_sk(const void* A_bf16, int64_t lda, int64_t rowsA_total, int64_t rA1, const void* B_planes, int64_t rowsB_total, int64_t rB1,
                                     void* C, int c_bf16, int64_t ldc_m, int64_t sC1, int nb1, int M, int N, int K, const float* scale, int scale_div,
                                     int64_t scale_bs, const float* bias, int64_t bias_bs, int act, void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_bf16_rows_impl("cti_gemm_bf16_rows_sk", A_bf16, lda, rowsA_total, rA1, B_planes, rowsB_total, rB1, C, c_bf16, ldc_m, sC1, nb1, M, N, K, scale, scale_div,
                               scale_bs, bias, bias_bs, act, workspace, workspace_bytes, stream);
}

extern "C" size_t cti_gemm_nt_workspace_bytes(int64_t rowsA_total, int64_t rowsB_total, int K, int prec) {
    if (prec == CTI_PREC_F32 || rowsA_total <= 0 || rowsB_total <= 0 || K <= 0) return 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t n = al(planes_bytes(rowsA_total + PLANE_SLACK_ROWS, K)) + al(planes_bytes(rowsB_total + PLANE_SLACK_ROWS, K));
    if (rowsA_total < (1ll << 31) && rowsB_total < (1ll << 31)) {           // split-K partials of the unbatched call (M, N = all rows)
        const int S = plan_ksplit((int)rowsA_total, (int)rowsB_total, planes_kp(K), 1);
        if (S > 1) n += al(sizeof(float) * (size_t)S * (size_t)rowsA_total * (size_t)rowsB_total);
    }
    return n;
}

extern "C" int cti_gemm_nt(const float* A, int64_t lda, int64_t rowsA_total, int64_t rA1, int64_t rA2, const float* B, int64_t ldb,
                           int64_t rowsB_total, int64_t rB1, int64_t rB2, float* C, int64_t ldc_m, int64_t ldc_n, int64_t sC1, int64_t sC2,
                           int nb1, int nb2, int M, int N, int K, const float* scale, int scale_div, int64_t scale_bs, const float* bias,
                           int64_t bias_bs, int act, int prec, void* workspace, size_t workspace_bytes, void* stream) {
    // A is ONE row-major matrix of rowsA_total rows x K (row stride lda); batch (b1,b2) uses rows [b1*rA1 + b2*rA2, +M).  Same for B / N.
    CTI_REQUIRE_PTR(A); CTI_REQUIRE_PTR(B); CTI_REQUIRE_PTR(C);
    CTI_REQUIRE(M > 0 && N > 0 && K > 0 && nb1 > 0 && nb2 > 0 && lda >= K && ldb >= K, CTI_E_SHAPE, "cti_gemm_nt: M=%d N=%d K=%d nb=%dx%d", M, N, K, nb1, nb2);
    CTI_REQUIRE((int64_t)(nb1 - 1) * rA1 + (int64_t)(nb2 - 1) * rA2 + M <= rowsA_total && (int64_t)(nb1 - 1) * rB1 + (int64_t)(nb2 - 1) * rB2 + N <= rowsB_total,
                CTI_E_SHAPE, "cti_gemm_nt: batches run past the operand rows");
    CTI_REQUIRE(act == CTI_ACT_NONE || act == CTI_ACT_RELU, CTI_E_UNSUPPORTED, "cti_gemm_nt: act=%d", act);
    if (prec == CTI_PREC_F32) {
        GemmP p{};
        p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc_m = ldc_m; p.ldc_n = ldc_n;
        p.sA1 = rA1 * lda; p.sA2 = rA2 * lda; p.sB1 = rB1 * ldb; p.sB2 = rB2 * ldb; p.sC1 = sC1; p.sC2 = sC2;
        p.nb1 = nb1; p.nb2 = nb2; p.M = M; p.N = N; p.K = K;
        p.scale = scale; p.scale_div = scale ? scale_div : 1; p.bias = bias; p.relu = act == CTI_ACT_RELU;
        p.scale_bs = scale_bs; p.bias_bs = bias_bs;
        return gemm_nt_f32(p, as_stream(stream));
    }
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_gemm_nt: prec=%d", prec);
    CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(workspace_bytes >= cti_gemm_nt_workspace_bytes(rowsA_total, rowsB_total, K, prec), CTI_E_WORKSPACE, "cti_gemm_nt: workspace too small");
    const int Kp = planes_kp(K);
    const int64_t ra = rowsA_total + PLANE_SLACK_ROWS, rb = rowsB_total + PLANE_SLACK_ROWS;
    unsigned short* ah = static_cast<unsigned short*>(workspace);
    unsigned short* al = ah + (size_t)ra * Kp;
    unsigned short* bh = reinterpret_cast<unsigned short*>(static_cast<char*>(workspace) + ((planes_bytes(ra, K) + 255) & ~(size_t)255));
    unsigned short* bl = bh + (size_t)rb * Kp;
    int rc = split_planes(A, lda, rowsA_total, K, ah, al, ra, as_stream(stream)); if (rc) return rc;
    rc = split_planes(B, ldb, rowsB_total, K, bh, bl, rb, as_stream(stream)); if (rc) return rc;
    PlaneGemmArgs g{};
    g.Ah = ah; g.Al = al; g.Bh = bh; g.Bl = bl; g.rows_allocA = ra; g.rows_allocB = rb;
    g.rA1 = rA1; g.rA2 = rA2; g.rB1 = rB1; g.rB2 = rB2; g.nb1 = nb1; g.nb2 = nb2;
    g.M = M; g.N = N; g.Kp = Kp; g.terms = prec == CTI_PREC_BF16X3 ? 3 : 1;
    g.C = C; g.ldc_m = ldc_m; g.ldc_n = ldc_n; g.sC1 = sC1; g.sC2 = sC2;
    g.epi = (ldc_n == 1) ? 0 : 3; g.gdiv = 1;
    g.scale = scale; g.scale_div = scale ? scale_div : 1; g.bias = bias; g.relu = act == CTI_ACT_RELU;
    g.scale_bs = scale_bs; g.bias_bs = bias_bs;
    if (nb1 == 1 && nb2 == 1 && M == rowsA_total && N == rowsB_total && ldc_n == 1) {
        const int S = plan_ksplit(M, N, Kp, 1);
        if (S > 1) {
            g.ksplit = S;
            g.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + ((planes_bytes(ra, K) + 255) & ~(size_t)255) +
                                                 ((planes_bytes(rb, K) + 255) & ~(size_t)255));
        }
    }
    return gemm_nt_planes(g, as_stream(stream));
}
