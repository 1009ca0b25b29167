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
#ifndef CTI_RN_ABL
#define CTI_RN_ABL 0            // timing-only ablation (wrong results): 1 = no mask loads (every kernel takes an all-ones mask)
#endif
__device__ __forceinline__ uchar4 ldm4(const uint8_t* p) { if (CTI_RN_ABL & 1) return make_uchar4(1, 1, 1, 1); return *reinterpret_cast<const uchar4*>(p); }
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

// ---- forward on the bf16 matrix cores (round 3; h = 512, hr = 16: the models' widths; every mode but exact fp32) ------------------------------------
// The fp32-MFMA forward above is bound by its own issue rate (16x16x4 f32: 1/16 of the bf16 rate -- 35 us of MFMA time alone for the visual branch, 186 us
// measured).  Here the products are bf16 hi / lo split products on the 16x16x16 MFMA (three per pair: fp32-grade, as everywhere else):
//   * the wave's x fragments are split ONCE into MFMA-ready hi / lo operands (2 + 2 registers per 16-deep K step: the same 128 registers the fp32 form
//     spends on raw x) and masked per rank with two v_perm_b32 + four v_and_b32 per step (mask bytes x 255 -> byte masks -> 16-bit lane masks);
//   * W_r arrives PRE-SPLIT (bf16 hi and lo planes written by rn_w_split_kernel into the caller's workspace, 1 MB) through LDS-DMA into a double-buffered
//     [plane][16 rows][1 056 B] image shared by the workgroup's four row tiles: 4x less L2 traffic than per-wave global fragments (590 MB per launch
//     otherwise), rows 264 dwords apart so that the 8-B fragment reads of 16 rows x 4 K quarters take the minimum two LDS passes.
typedef short rn_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned rn_u32x2 __attribute__((ext_vector_type(2)));
constexpr int RNM_PITCH = 1056;                                      // bytes between W rows in LDS (h = 512 bf16 + 32)
constexpr int RNM_BUF = 2 * 16 * RNM_PITCH;                          // one rank: hi plane + lo plane

__global__ __launch_bounds__(256) void rn_w_split_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wh, unsigned short* __restrict__ Wl, int64_t n, int h) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 w = ld4(W + i);
    const float v[4] = {w.x, w.y, w.z, w.w};
    unsigned short hb[4], lb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = static_cast<__bf16>(v[e]);
        hb[e] = __builtin_bit_cast(unsigned short, hh);
        lb[e] = __builtin_bit_cast(unsigned short, static_cast<__bf16>(v[e] - static_cast<float>(hh)));
    }
    // K order of the planes: inside every 64-deep group the 4-element pieces are stored (i', kq)-major instead of (kq, i')-major -- piece (kq, i') = elements
    // 16 kq + 4 i' .. + 3 of the group goes to position 16 i' + 4 kq -- so that MFMA step s = 4 j + i' finds the four K quarters' pieces side by side (the
    // conflict-free LDS read) while lane quarter kq multiplies x[64 j + 16 kq + 4 i' ..]: its FOUR steps' mask bytes are then one contiguous 16-B load
    const int64_t row = i / h;
    const int k0 = (int)(i - row * h), grp = k0 >> 6, rem = k0 & 63, kq = rem >> 4, ip = (rem & 15) >> 2;
    const int64_t o = row * h + grp * 64 + ip * 16 + kq * 4;
    *reinterpret_cast<uint2*>(Wh + o) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
    *reinterpret_cast<uint2*>(Wl + o) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
}

template <int TERMS>
__global__ __launch_bounds__(256) void rn_fwd_mfma_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, const unsigned short* __restrict__ Wh,
                                                          const unsigned short* __restrict__ Wl, const float* __restrict__ scale, const float* __restrict__ bias,
                                                          float* __restrict__ y, int64_t rows, int R, float inv_keep, int relu, int r_per) {
    constexpr int h = 512, hr = 16, NS = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 buffers][hi | lo][16 rows][RNM_PITCH]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wid) * 16;
    const int64_t m = tile0 + l15;                                   // the row this lane feeds as the A operand
    const bool mok = m < rows;                                       // (no early exit: every wave takes part in the DMA and the barriers)
    const int64_t mc = mok ? m : rows - 1;
    rn_u32x2 xh[NS], xl[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float4 v4 = ld4(x + mc * h + (s >> 2) * 64 + 16 * kq + 4 * (s & 3));       // step s = 4 j + i': this lane's K quartet is 64 j + 16 kq + 4 i' ..
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        unsigned hb[4], lb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 hh = static_cast<__bf16>(v[e]);
            hb[e] = __builtin_bit_cast(unsigned short, hh);
            lb[e] = TERMS == 3 ? (unsigned)__builtin_bit_cast(unsigned short, static_cast<__bf16>(v[e] - static_cast<float>(hh))) : 0u;
        }
        xh[s] = rn_u32x2{hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
        xl[s] = rn_u32x2{lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
    }
    const int r_lo = blockIdx.y * r_per, r_hi = min(R, r_lo + r_per);
    // W_r: 32 pieces of 1 KiB (plane, row), eight per wave, one LDS-DMA instruction each (64 lanes x 16 B = one 512-element row of a plane)
    auto dma_w = [&](int r, int buf) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = wid * 8 + u, plane = id >> 4, row = id & 15;
            if (TERMS == 1 && plane) continue;
            const unsigned short* src = (plane ? Wl : Wh) + ((int64_t)r * hr + row) * h + lane * 8;
            char* dst = smem + buf * RNM_BUF + (plane * 16 + row) * RNM_PITCH;              // wave-uniform; the hardware adds lane * 16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    dma_w(r_lo, 0);
    const int ldy = R * hr;
    for (int r = r_lo; r < r_hi; ++r) {
        const int buf = (r - r_lo) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's pieces of W_r have landed
        __builtin_amdgcn_s_barrier();                                    // everyone's have; and everyone is done with the other buffer (raw: __syncthreads() is a fence too)
        const uint8_t* mp = mask + ((int64_t)r * rows + mc) * h + 16 * kq;
        const char* wb = smem + buf * RNM_BUF + l15 * RNM_PITCH + kq * 8;
        rf32x4 acc = {0.f, 0.f, 0.f, 0.f};
        uint4 mq[NS / 4];                                                // requested before the next rank's pieces are issued
#pragma unroll
        for (int j = 0; j < NS / 4; ++j) mq[j] = (CTI_RN_ABL & 1) ? make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u) : *reinterpret_cast<const uint4*>(mp + j * 64);       // the mask bytes of four steps per 16-B load
        if (r + 1 < r_hi) dma_w(r + 1, buf ^ 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned mq4[4] = {mq[s >> 2].x, mq[s >> 2].y, mq[s >> 2].z, mq[s >> 2].w};
            const unsigned mk = mq4[s & 3];                                                  // four mask bytes (0 / 1) of this lane's k quartet
            const unsigned t = mk * 255u;                                                    // 0x00 / 0xFF per byte
            const unsigned m01 = __builtin_amdgcn_perm(t, t, 0x01010000u), m23 = __builtin_amdgcn_perm(t, t, 0x03030202u);
            const rn_u32x2 ahu = {xh[s][0] & m01, xh[s][1] & m23};
            const rn_s16x4 ah = __builtin_bit_cast(rn_s16x4, ahu);
            const rn_s16x4 bh = __builtin_bit_cast(rn_s16x4, *reinterpret_cast<const rn_u32x2*>(wb + s * 32));
            if (TERMS == 3) {
                const rn_u32x2 alu = {xl[s][0] & m01, xl[s][1] & m23};
                const rn_s16x4 al = __builtin_bit_cast(rn_s16x4, alu);
                const rn_s16x4 bl = __builtin_bit_cast(rn_s16x4, *reinterpret_cast<const rn_u32x2*>(wb + 16 * RNM_PITCH + s * 32));
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
        }
        {                                                                // C/D: column = lane & 15 (n), rows 4 * (lane >> 4) + i
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

// (Round 3, built, parity-green and removed: the INPUT gradient the same way -- D^T[k, m] = W_r^T dzs_r^T per 16 columns on the 16x16x16 MFMA, the masked sum over
// the ranks in registers, W_r^T through a four-slot LDS-DMA ring -- measured 70 us per branch against the fp32-MFMA kernel's 46: a rank's arithmetic (~1 000
// cycles) is shorter than one DMA latency, and every register load the compiler can see inside the rank loop (mask bytes, the dzs fragment) makes it wait for
// vmcnt(0) at the first use, i.e. for the prefetches issued in between; hiding the loads in inline asm with hand-counted waits moved nothing in the forward
// kernel either (43.0 against 40.7 us).  The mask traffic itself is 7-8 us of each of these kernels (CTI_RN_ABL=1).)

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

// The same forward on the bf16 matrix cores (rn_fwd_mfma_kernel): h = 512, hr = 16, prec = CTI_PREC_BF16X3 (three split products per pair) or CTI_PREC_BF16 (one).
// workspace: cti_ranknets_drop_fwd_mfma_workspace_bytes(h, R, hr) bytes for the bf16 hi / lo planes of W.  CTI_E_UNSUPPORTED (nothing launched, no message)
// for other widths, the exact-fp32 mode or unaligned operands: the caller takes cti_ranknets_drop_fwd.
extern "C" size_t cti_ranknets_drop_fwd_mfma_workspace_bytes(int h, int R, int hr) {
    if (h <= 0 || R <= 0 || hr <= 0) return 0;
    return 2 * sizeof(unsigned short) * (size_t)R * hr * h + 256;
}

extern "C" int cti_ranknets_drop_fwd_mfma(const float* x, const uint8_t* mask, const float* W, const float* scale, const float* bias, float* y,
                                          int64_t rows, int h, int R, int hr, float p, int relu, int prec, void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(mask); CTI_REQUIRE_PTR(W); CTI_REQUIRE_PTR(scale); CTI_REQUIRE_PTR(y);
    CTI_REQUIRE(rows > 0 && h > 0 && R > 0 && hr > 0 && p >= 0.f && p < 1.f, CTI_E_SHAPE, "cti_ranknets_drop_fwd_mfma: rows=%lld h=%d R=%d hr=%d p=%f",
                (long long)rows, h, R, hr, p);
    if (prec != CTI_PREC_BF16X3 && prec != CTI_PREC_BF16) return CTI_E_UNSUPPORTED;
    if (h != 512 || hr != 16 || R > 65535 || (rows + 63) / 64 > 0x7fffffffLL || !aligned16(x) || !aligned16(W) || (reinterpret_cast<uintptr_t>(mask) & 15))
        return CTI_E_UNSUPPORTED;
    CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(workspace_bytes >= cti_ranknets_drop_fwd_mfma_workspace_bytes(h, R, hr) && aligned16(workspace), CTI_E_WORKSPACE,
                "cti_ranknets_drop_fwd_mfma: workspace too small or not 16-B aligned");
    const int64_t nW = (int64_t)R * hr * h;
    unsigned short* Wh = static_cast<unsigned short*>(workspace);
    unsigned short* Wl = Wh + nW;
    hipLaunchKernelGGL(rn_w_split_kernel, dim3((unsigned)((nW / 4 + 255) / 256)), dim3(256), 0, as_stream(stream), W, Wh, Wl, nW, h);
    int rc = launch_status("cti_ranknets_drop_fwd_mfma/split"); if (rc) return rc;
    const int64_t wgs = (rows + 63) / 64;
    int rs = 1;                                                      // split the ranks over workgroups while they all stay co-resident (two per CU: 67 KB of LDS each)
    static const long long rn_wgs = [] { const char* e = getenv("CTI_RN_MFMA_WGS"); return e ? atoll(e) : 512LL; }();      // A/B knob
    while (rs < R && wgs * rs * 2 <= rn_wgs) rs *= 2;
    const int r_per = (R + rs - 1) / rs;
    const dim3 grid((unsigned)wgs, (unsigned)((R + r_per - 1) / r_per));
    const float inv_keep = 1.f / (1.f - p);
    const size_t lds = 2 * (size_t)RNM_BUF;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rn_fwd_mfma_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(rn_fwd_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "cti_ranknets_drop_fwd_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    if (prec == CTI_PREC_BF16X3)
        hipLaunchKernelGGL(rn_fwd_mfma_kernel<3>, grid, dim3(256), lds, as_stream(stream), x, mask, Wh, Wl, scale, bias, y, rows, R, inv_keep, relu, r_per);
    else
        hipLaunchKernelGGL(rn_fwd_mfma_kernel<1>, grid, dim3(256), lds, as_stream(stream), x, mask, Wh, Wl, scale, bias, y, rows, R, inv_keep, relu, r_per);
    return launch_status("cti_ranknets_drop_fwd_mfma");
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
