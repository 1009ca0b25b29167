// cti_ranknets.hip -- the R rank nets of TCNet in TRAIN mode (src/tc.py:29-31, 44-46: R x FCNet([h, h/R]) on one shared input, each
// FCNet with its own nn.Dropout in front of its weight-normed Linear, src/fc.py:25-28).  R independent masks of the same (rows, h)
// input: the general route materialises the R masked copies (R*rows*h floats: 604 MB for the visual branch at B = 256), transposes and
// plane-splits them for three batched GEMMs.  Here the mask is applied where the operand fragment is formed, so the copies never exist:
//   forward   y[m, r*hr+n]  = act(scale_r/(1-p) * sum_k x[m,k] mask[r,m,k] W[r*hr+n,k] + bias)         x fragments stay in registers over r
//   weights   G[r*hr+n, k]  = 1/(1-p) * sum_m dzs[m, r*hr+n] x[m,k] mask[r,m,k]                         contraction over the rows
//   input     dx[m, k]      = 1/(1-p) * sum_r mask[r,m,k] * (sum_n dzs[m, r*hr+n] W[r*hr+n,k])         mask applied to each rank's tile
// All three run on the fp32 MFMA (v_mfma_f32_16x16x4_f32: one f32 per lane and operand, exact fp32 products, so every precision mode
// shares them).  Its fragment is "A[row = lane&15][k = lane>>4]": a lane loads FOUR consecutive elements of the contraction axis (forward,
// input gradient) or of the output-column axis (weights, input gradient) with one 16-B load and feeds element j to MFMA j -- the
// contraction order, resp. the output column order, is permuted consistently on both operands, and every global access is a 64-B run per
// 16 lanes.  HBM traffic per branch: the mask (R*rows*h bytes) once per kernel + x / dzs / W; nothing else.
#include "cti_common.h"
#include <cstdlib>

namespace cti {
namespace {

typedef float rf32x4 __attribute__((ext_vector_type(4)));

// Every load below is UNCONDITIONAL from a clamped (always valid) address; out-of-range lanes are zeroed by a select on the value, or
// left as garbage where the result is never stored (a predicated load costs an exec-mask region and a wait of its own per instruction).
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uchar4 ldm4(const uint8_t* p) { return *reinterpret_cast<const uchar4*>(p); }
__device__ __forceinline__ float4 ld4p(const float* p, bool ok) { return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ uchar4 ldm4p(const uint8_t* p, bool ok) { return ok ? *reinterpret_cast<const uchar4*>(p) : make_uchar4(0, 0, 0, 0); }
__device__ __forceinline__ float4 sel4(bool ok, float4 v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); }

// ---- forward: a wave owns 16 rows (its x fragments: NS float4 per lane) and walks its share of the ranks ---------------------------
// 186 us for the visual branch (rows 9216, h 512, R 32), ~6x its MFMA time, and remarkably indifferent to what was tried on it
// (tools/abl_ranknets.py, each measured): W_r staged through LDS by the workgroup, the mask as 16-B loads in a permuted k order, all
// loads of a rank batched unconditionally at its top, a 4-deep prefetch ring, two or four accumulator chains -- 183-195 us every time;
// unconditional loads everywhere let the scheduler hoist 64 loads per rank: 256 VGPRs, one wave per SIMD, 423 us.  Ablations: no MFMAs
// 153 us, no mask loads 142 us, no W reads 187 us, no x loads 183 us, no stores 185 us, the bare loops 45 us.  The x fragments (128 VGPRs)
// cap it at 2-3 waves per SIMD; the next design would hold them in LDS or split K over waves.
template <int NS>
__global__ __launch_bounds__(256) void rn_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, const float* __restrict__ W,
                                                     const float* __restrict__ scale, const float* __restrict__ bias, float* __restrict__ y,
                                                     int64_t rows, int h, int R, int hr, float inv_keep, int relu, int r_per) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wid) * 16;
    if (tile0 >= rows) return;
    const int64_t m = tile0 + l15;                                   // the row this lane feeds as the A operand
    const bool mok = m < rows, nok = l15 < hr;
    float4 xf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) xf[s] = ld4p(x + m * h + s * 16 + 4 * kq, mok && s * 16 + 4 * kq < h);
    const int r_lo = blockIdx.y * r_per, r_hi = min(R, r_lo + r_per);
    const int ldy = R * hr;
    for (int r = r_lo; r < r_hi; ++r) {
        const uint8_t* mp = mask + ((int64_t)r * rows + (mok ? m : 0)) * h + 4 * kq;
        const float* wp = W + ((int64_t)r * hr + (nok ? l15 : 0)) * h + 4 * kq;
        rf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s * 16 < h) {                                        // uniform
                const bool kok = s * 16 + 4 * kq < h;
                const uchar4 mk = ldm4p(mp + s * 16, mok && kok);
                const float4 w = ld4p(wp + s * 16, nok && kok);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mk.x ? xf[s].x : 0.f, w.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mk.y ? xf[s].y : 0.f, w.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mk.z ? xf[s].z : 0.f, w.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mk.w ? xf[s].w : 0.f, w.w, acc, 0, 0, 0);
            }
        }
        if (nok) {                                                   // C/D: column = lane&15 (n), rows 4*(lane>>4) + i
            const float sc = scale[r] * inv_keep, bb = bias ? bias[r * hr + l15] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = tile0 + 4 * kq + i;
                if (row < rows) {
                    float v = fmaf(acc[i], sc, bb);
                    if (relu) v = relu_nan(v);
                    y[row * ldy + r * hr + l15] = v;
                }
            }
        }
    }
}

// ---- weight gradient: one workgroup per (64 input columns, rank); its 16 waves split the rows and meet in LDS in a fixed order ---------
__global__ __launch_bounds__(1024) void rn_dw_kernel(const float* __restrict__ dzs, const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                     float* __restrict__ G, int64_t rows, int h, int R, int hr, float inv_keep, int64_t rows_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [16 waves][16 values][64 lanes]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int k0 = blockIdx.x * 64, r = blockIdx.y;
    const int k = k0 + 4 * l15;
    const bool kok = k < h, nok = l15 < hr;
    const int ldz = R * hr;
    const int64_t m_lo = (int64_t)wid * rows_per_wave, m_hi = min(rows, m_lo + rows_per_wave);
    rf32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = rf32x4{0.f, 0.f, 0.f, 0.f};
    const int kc = kok ? k : 0;                                      // past-the-end columns are computed on column 0's data and never stored
    const float* zp = dzs + (int64_t)r * hr + (nok ? l15 : 0);
    const float* xp = x + kc;
    const uint8_t* mp = mask + (int64_t)r * rows * h + kc;
    for (int64_t m0 = m_lo; m0 < m_hi; m0 += 8) {                    // two 4-row steps per trip: their loads are in flight together
        const int64_t ra = m0 + kq, rb = m0 + 4 + kq;
        const bool oa = ra < m_hi, ob = rb < m_hi;
        const int64_t rac = oa ? ra : rows - 1, rbc = ob ? rb : rows - 1;
        const float za = zp[rac * ldz], zb = zp[rbc * ldz];
        const float a0 = (oa && nok) ? za : 0.f, a1 = (ob && nok) ? zb : 0.f;
        const float4 x0 = sel4(oa, ld4(xp + rac * h)), x1 = sel4(ob, ld4(xp + rbc * h));
        const uchar4 k0m = ldm4(mp + rac * h), k1m = ldm4(mp + rbc * h);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, k0m.x ? x0.x : 0.f, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, k0m.y ? x0.y : 0.f, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, k0m.z ? x0.z : 0.f, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, k0m.w ? x0.w : 0.f, acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, k1m.x ? x1.x : 0.f, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, k1m.y ? x1.y : 0.f, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, k1m.z ? x1.z : 0.f, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, k1m.w ? x1.w : 0.f, acc[3], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) sm[(wid * 16 + t * 4 + i) * 64 + lane] = acc[t][i];
    __syncthreads();
    {                                                                // thread e = (t, i, lane'): acc[t][i] of lane' = G[r*hr + 4*kq' + i][k0 + 4*l15' + t]
        const int e = threadIdx.x, ln = e & 63, ti = e >> 6, t = ti >> 2, i = ti & 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += sm[w * 1024 + e];
        const int n = 4 * (ln >> 4) + i, kk = k0 + 4 * (ln & 15) + t;
        if (n < hr && kk < h) G[((int64_t)r * hr + n) * h + kk] = s * inv_keep;
    }
}

// ---- input gradient: a workgroup owns a 16-row x 64-column tile of dx; its 4 waves sum the masked products of a quarter of the ranks each ---
__global__ __launch_bounds__(256) void rn_dx_kernel(const float* __restrict__ dzs, const float* __restrict__ W, const uint8_t* __restrict__ mask,
                                                    float* __restrict__ dx, int64_t rows, int h, int R, int hr, float inv_keep, int z4ok) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int64_t tile0 = (int64_t)blockIdx.x * 16;                  // one 16-row tile per workgroup; its 4 waves take a quarter of the ranks each
    const int k = blockIdx.y * 64 + 4 * l15;
    const bool kok = k < h;
    const int64_t ma = tile0 + l15;                                  // A operand row
    const bool mok = ma < rows;
    const int ldz = R * hr;
    rf32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = rf32x4{0.f, 0.f, 0.f, 0.f};
    const float* zrow = dzs + (mok ? ma : rows - 1) * ldz + 4 * kq;
    const int kc = kok ? k : 0;                                      // past-the-end columns / rows: computed on valid data, never stored
    int64_t mrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mrow[i] = min(tile0 + 4 * kq + i, rows - 1) * h + kc;
    const int r_per = (R + 3) / 4, r_lo = wid * r_per, r_hi = min(R, r_lo + r_per);
    for (int r = r_lo; r < r_hi; ++r) {
        // A element j <-> n = 4*kq + j (the lane's contraction slot kq of MFMA j); B element: W[r*hr + 4*kq + j][k .. k+3]
        float a[4];
        if (z4ok && 4 * kq + 4 <= hr) {
            const float4 z = sel4(mok, ld4(zrow + r * hr));
            a[0] = z.x; a[1] = z.y; a[2] = z.z; a[3] = z.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float z = zrow[r * hr + (4 * kq + j < hr ? j : -4 * kq)]; a[j] = (mok && 4 * kq + j < hr) ? z : 0.f; }
        }
        rf32x4 P[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) P[t] = rf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 w = ld4(W + ((int64_t)r * hr + (4 * kq + j < hr ? 4 * kq + j : 0)) * h + kc);     // rows past hr meet a[j] = 0
            P[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], w.x, P[0], 0, 0, 0);
            P[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], w.y, P[1], 0, 0, 0);
            P[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], w.z, P[2], 0, 0, 0);
            P[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], w.w, P[3], 0, 0, 0);
        }
        const uint8_t* mp = mask + (int64_t)r * rows * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                               // P[t][i] = product at (row tile0 + 4*kq + i, column k + t)
            const uchar4 mk = ldm4(mp + mrow[i]);
            acc[0][i] += mk.x ? P[0][i] : 0.f;
            acc[1][i] += mk.y ? P[1][i] : 0.f;
            acc[2][i] += mk.z ? P[2][i] : 0.f;
            acc[3][i] += mk.w ? P[3][i] : 0.f;
        }
    }
    if (wid > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wid - 1][t * 4 + i][lane] = acc[t][i];
    }
    __syncthreads();
    if (wid == 0 && kok) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] = ((acc[t][i] + red[0][t * 4 + i][lane]) + red[1][t * 4 + i][lane]) + red[2][t * 4 + i][lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = tile0 + 4 * kq + i;
            if (row < rows)
                *reinterpret_cast<float4*>(dx + row * h + k) = make_float4(acc[0][i] * inv_keep, acc[1][i] * inv_keep, acc[2][i] * inv_keep, acc[3][i] * inv_keep);
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool rn_shape_ok(int64_t rows, int h, int R, int hr) {
    return rows > 0 && h > 0 && R > 0 && hr > 0 && hr <= 16 && h % 4 == 0 && R <= 65535 && (rows + 15) / 16 <= 0x7fffffffLL;
}

}  // namespace
}  // namespace cti

using namespace cti;

extern "C" int cti_ranknets_drop_fwd(const float* x, const uint8_t* mask, const float* W, const float* scale, const float* bias, float* y,
                                     int64_t rows, int h, int R, int hr, float p, int relu, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(mask); CTI_REQUIRE_PTR(W); CTI_REQUIRE_PTR(scale); CTI_REQUIRE_PTR(y);
    CTI_REQUIRE(rows > 0 && h > 0 && R > 0 && hr > 0 && p >= 0.f && p < 1.f, CTI_E_SHAPE, "cti_ranknets_drop_fwd: rows=%lld h=%d R=%d hr=%d p=%f",
                (long long)rows, h, R, hr, p);
    if (!rn_shape_ok(rows, h, R, hr) || h > 512 || !aligned16(x) || !aligned16(W) || (reinterpret_cast<uintptr_t>(mask) & 3)) return CTI_E_UNSUPPORTED;
    const int64_t wgs = (rows + 63) / 64;
    int rs = 1;                                                      // split the ranks until ~2 waves per SIMD are in flight
    static const long long rn_target = [] { const char* e = getenv("CTI_RN_WAVES"); return e ? atoll(e) : 4096LL; }();   // A/B knob; 2048 -> 4096 waves: 192 -> 169 us at rows 9216
    while (rs < R && wgs * 4 * rs < rn_target) rs *= 2;
    const int r_per = (R + rs - 1) / rs;
    const dim3 grid((unsigned)wgs, (unsigned)((R + r_per - 1) / r_per));
    const float inv_keep = 1.f / (1.f - p);
#define CTI_RNF(NSv) hipLaunchKernelGGL((rn_fwd_kernel<NSv>), grid, dim3(256), 0, as_stream(stream), x, mask, W, scale, bias, y, rows, h, R, hr, inv_keep, relu, r_per)
    if (h <= 64) CTI_RNF(4); else if (h <= 128) CTI_RNF(8); else if (h <= 256) CTI_RNF(16); else CTI_RNF(32);
#undef CTI_RNF
    return launch_status("cti_ranknets_drop_fwd");
}

extern "C" int cti_ranknets_drop_dw(const float* dzs, const float* x, const uint8_t* mask, float* G, int64_t rows, int h, int R, int hr, float p,
                                    void* stream) {
    CTI_REQUIRE_PTR(dzs); CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(mask); CTI_REQUIRE_PTR(G);
    CTI_REQUIRE(rows > 0 && h > 0 && R > 0 && hr > 0 && p >= 0.f && p < 1.f, CTI_E_SHAPE, "cti_ranknets_drop_dw: rows=%lld h=%d R=%d hr=%d p=%f",
                (long long)rows, h, R, hr, p);
    if (!rn_shape_ok(rows, h, R, hr) || !aligned16(x) || (reinterpret_cast<uintptr_t>(mask) & 3)) return CTI_E_UNSUPPORTED;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rn_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return fail((int)e, "cti_ranknets_drop_dw: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    const int64_t rpw = ((rows + 15) / 16 + 7) / 8 * 8;
    hipLaunchKernelGGL(rn_dw_kernel, dim3((h + 63) / 64, R), dim3(1024), 64 * 1024, as_stream(stream), dzs, x, mask, G, rows, h, R, hr, 1.f / (1.f - p), rpw);
    return launch_status("cti_ranknets_drop_dw");
}

extern "C" int cti_ranknets_drop_dx(const float* dzs, const float* W, const uint8_t* mask, float* dx, int64_t rows, int h, int R, int hr, float p,
                                    void* stream) {
    CTI_REQUIRE_PTR(dzs); CTI_REQUIRE_PTR(W); CTI_REQUIRE_PTR(mask); CTI_REQUIRE_PTR(dx);
    CTI_REQUIRE(rows > 0 && h > 0 && R > 0 && hr > 0 && p >= 0.f && p < 1.f, CTI_E_SHAPE, "cti_ranknets_drop_dx: rows=%lld h=%d R=%d hr=%d p=%f",
                (long long)rows, h, R, hr, p);
    if (!rn_shape_ok(rows, h, R, hr) || !aligned16(W) || !aligned16(dx) || (reinterpret_cast<uintptr_t>(mask) & 3)) return CTI_E_UNSUPPORTED;
    const int z4ok = hr % 4 == 0 && aligned16(dzs);
    hipLaunchKernelGGL(rn_dx_kernel, dim3((unsigned)((rows + 15) / 16), (h + 63) / 64), dim3(256), 0, as_stream(stream), dzs, W, mask, dx, rows, h, R, hr,
                       1.f / (1.f - p), z4ok);
    return launch_status("cti_ranknets_drop_dx");
}
