// cti_gemm_bf16x3.hip -- fp32-grade NT GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16).
//
// gfx950 has no TF32/xf32; its exact-fp32 MFMA runs at 1/16 of the bf16 rate.  Here every fp32 operand x is split
// once into two bf16 planes, hi = bf16(x) and lo = bf16(x - hi) (|x - hi - lo| <= 2^-17 |x|), and each product is
// issued as three bf16 MFMAs into one fp32 accumulator:  a*b ~= ah*bh + ah*bl + al*bh  (the dropped al*bl term is
// <= 2^-16 relative).  Measured against the float64 oracle at the BASELINE config-1 shapes the whole TCNet.forward
// stays at 1e-5 normalised max error (tolerance 1e-4), at 3/16 of the exact-fp32 MFMA issue cost.
//
// Stage 1 (split kernel): fp32 [rows, K] (row stride ld) -> hi/lo planes [rows, Kp], Kp = K rounded up to BK,
//   zero-filled tail, so that the GEMM loads 16-B chunks with no K predicate.
// Stage 2 (GEMM): 128 x 128 x 32 tile per 256-thread workgroup, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 tiles of
//   32 x 32.  Planes are staged by LDS-DMA (global_load_lds_dwordx4: 16 B per lane straight into LDS, no VGPRs),
//   two LDS buffers, next tile's DMA issued before the current tile's MFMAs, one barrier per K-step.  The LDS image
//   is lane-linear ([row][4 x 16 B]); the bank-conflict-free layout for ds_read_b128 comes from permuting the SOURCE
//   chunk (c' = c ^ ((row >> 2) & 3)) and applying the same XOR on the read.
#include "cti_common.h"

namespace cti {

namespace {

constexpr int BN = 128, BK = 32, NSTAGE = 3;
constexpr int ROW_BYTES = BK * 2;                         // one tile row of one plane: 64 B = 4 chunks of 16 B
constexpr int B_PLANE = BN * ROW_BYTES;                   // 8 KiB
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x)); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// ---- split: one thread per 8 consecutive k of one row ------------------------------------------------------------
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int K, int Kp,
                                                    unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
    const int cpr = Kp >> 3;                                       // 8-element chunks per row
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * cpr) return;
    const int64_t r = idx / cpr;
    const int k0 = (int)(idx - r * cpr) * 8;
    const float* src = x + r * ld + k0;
    float v[8];
    if (k0 + 8 <= K && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k0 + j < K) ? src[j] : 0.f;
    }
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = bf16_bits(v[j]);
        l[j] = bf16_bits(v[j] - bf16_to_f32(h[j]));
    }
    uint4 ph, pl;
    ph.x = h[0] | ((unsigned)h[1] << 16); ph.y = h[2] | ((unsigned)h[3] << 16); ph.z = h[4] | ((unsigned)h[5] << 16); ph.w = h[6] | ((unsigned)h[7] << 16);
    pl.x = l[0] | ((unsigned)l[1] << 16); pl.y = l[2] | ((unsigned)l[3] << 16); pl.z = l[4] | ((unsigned)l[5] << 16); pl.w = l[6] | ((unsigned)l[7] << 16);
    *reinterpret_cast<uint4*>(hi + r * Kp + k0) = ph;
    *reinterpret_cast<uint4*>(lo + r * Kp + k0) = pl;
}

struct PlaneGemmP {
    const unsigned short* Ah; const unsigned short* Al; const unsigned short* Bh; const unsigned short* Bl;
    float* C;
    int64_t lda, ldb;                          // plane row strides in ELEMENTS (multiples of 8)
    int64_t ldc_m, ldc_n;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;      // batch strides (elements)
    int nb2;
    int M, N, Kp;
    const float* scale; int scale_div; const float* bias; int relu;
    // EPI_PLANES: the result is written as bf16 hi/lo planes [row][ldp] (columns N..Np-1 zero-filled) instead of fp32
    unsigned short* Ph; unsigned short* Pl; int64_t ldp; int Np;
    // EPI_INTERLEAVE: GEMM row m' = m*gdiv + g addresses C[(m'/gdiv)*ldc_m + (m'%gdiv) + n*ldc_n] (gdiv = G, ldc_n = G)
    int gdiv;
};
enum { EPI_F32 = 0, EPI_PLANES = 1, EPI_INTERLEAVE2 = 2, EPI_INTERLEAVE = 3 };

// LDS-DMA of one ROWS x 32 bf16 plane tile (ROWS*4 chunks of 16 B, ROWS*4/NTHR per thread).  LDS position p (chunk
// index) = (row = p >> 2, c' = p & 3) receives source chunk c = c' ^ ((row >> 2) & 3) of that row.
template <int ROWS, int NTHR>
__device__ __forceinline__ void dma_plane(const unsigned short* __restrict__ g, int64_t ld, int k0, char* lds_plane, int t) {
#pragma unroll
    for (int i = 0; i < ROWS * 4 / NTHR; ++i) {
        const int p = t + NTHR * i;
        const int row = p >> 2, c = (p & 3) ^ ((row >> 2) & 3);
        const unsigned short* src = g + (int64_t)row * ld + k0 + c * 8;
        char* dst = lds_plane + ((t & ~63) + NTHR * i) * 16;          // wave-uniform base; the hardware adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 frag(const char* lds_plane, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(lds_plane + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WM = waves along M (2 or 4); the workgroup is WM x 2 waves, each owning a 64 x 64 sub-tile: tile (WM*64) x 128 x 32.
template <int TERMS, int EPI, int WM>
__global__ __launch_bounds__(WM * 128) void gemm_planes_kernel(PlaneGemmP p) {
    constexpr int BM = WM * 64, NTHR = WM * 128;
    constexpr int A_PLANE = BM * ROW_BYTES;
    constexpr int STAGE = 2 * (A_PLANE + B_PLANE);                       // [A_hi | A_lo | B_hi | B_lo]
    constexpr int NDMA = (TERMS == 3 ? 2 : 1) * (BM * 4 / NTHR + BN * 4 / NTHR);   // DMA instructions per thread per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];      // NSTAGE stages; reused by the staged epilogue
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = ((EPI == EPI_PLANES ? p.Np : p.N) + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    int z, tm, tn;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, z, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int b1 = z / p.nb2, b2 = z % p.nb2;
    const int64_t offA = b1 * p.sA1 + b2 * p.sA2 + (int64_t)m0 * p.lda;
    const int64_t offB = b1 * p.sB1 + b2 * p.sB2 + (int64_t)n0 * p.ldb;
    const unsigned short* Ah = p.Ah + offA; const unsigned short* Al = p.Al + offA;
    const unsigned short* Bh = p.Bh + offB; const unsigned short* Bl = p.Bl + offB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto stage = [&](int slot, int k0) {
        char* s = smem + slot * STAGE;
        dma_plane<BM, NTHR>(Ah, p.lda, k0, s, t);
        dma_plane<BN, NTHR>(Bh, p.ldb, k0, s + 2 * A_PLANE, t);
        if (TERMS == 3) {
            dma_plane<BM, NTHR>(Al, p.lda, k0, s + A_PLANE, t);
            dma_plane<BN, NTHR>(Bl, p.ldb, k0, s + 2 * A_PLANE + B_PLANE, t);
        }
    };

    // 3-slot LDS ring, two K-steps of LDS-DMA in flight behind the MFMAs.  Per step: a COUNTED vmcnt retires exactly the
    // oldest tile (this wave's share), the raw barrier makes every wave's share visible and proves that the slot about to
    // be refilled (read one step ago) is idle, then the refill is issued and the MFMAs run.  No vmcnt(0) in the loop.
    const int nk = p.Kp / BK;
    stage(0, 0);
    if (nk > 1) stage(1, BK);
    const int r = lane & 31, h = lane >> 5;
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage(slot == 0 ? 2 : slot - 1, (kt + 2) * BK);          // slot of step kt+2 == slot of step kt-1
        const char* s = smem + slot * STAGE;
        const char* sAh = s + (wm * 64) * ROW_BYTES;
        const char* sAl = s + A_PLANE + (wm * 64) * ROW_BYTES;
        const char* sBh = s + 2 * A_PLANE + (wn * 64) * ROW_BYTES;
        const char* sBl = s + 2 * A_PLANE + B_PLANE + (wn * 64) * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int c = 2 * ks + h;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = frag(sAh, i * 32 + r, c);
                bh[i] = frag(sBh, i * 32 + r, c);
                if (TERMS == 3) { al[i] = frag(sAl, i * 32 + r, c); bl[i] = frag(sBl, i * 32 + r, c); }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (TERMS == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    __syncthreads();                              // every wave is done reading the ring before the epilogue reuses it

    if (EPI == EPI_PLANES || EPI == EPI_F32) {
        // Staged epilogue.  Each wave parks its 64 x 64 fp32 sub-tile in its own 16 KiB of the (now idle) LDS
        // ([row][16 slots of 16 B], slot ^= row & 1: conflict-free for the ds_write_b32 column writes and for the
        // ds_read_b128 row reads), then every lane owns 8 consecutive columns of a row: scale/bias/ReLU, and either
        // two 16-B fp32 stores or one 16-B hi + one 16-B lo bf16 store -- 128/256 contiguous bytes per row.
        float* stg = reinterpret_cast<float*>(smem) + wid * 4096;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const int col = j * 32 + (lane & 31);
                    stg[row * 64 + ((((col >> 2) ^ (row & 1))) << 2) + (col & 3)] = acc[i][j][e];
                }
        __syncthreads();
        const int c8 = lane & 7;
        const int nb = n0 + wn * 64 + c8 * 8;                       // first of this lane's 8 columns
        const int ncols = (EPI == EPI_PLANES) ? p.Np : p.N;
        float sc[8], bi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = nb + u;
            const bool real = n < p.N;
            sc[u] = (real && p.scale) ? p.scale[n / p.scale_div] : 1.f;
            bi[u] = (real && p.bias) ? p.bias[n] : 0.f;
        }
        const int64_t boff = b1 * p.sC1 + b2 * p.sC2;
        const bool vecC = (EPI == EPI_F32) && (p.ldc_n == 1) && ((p.ldc_m & 3) == 0) && ((boff & 3) == 0) &&
                          ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int m = m0 + wm * 64 + row;
            const float4 x0 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * c8) ^ (row & 1)) << 2));
            const float4 x1 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * c8 + 1) ^ (row & 1)) << 2));
            if (m >= p.M || nb >= ncols) continue;
            float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = x[u] * sc[u] + bi[u];
                if (p.relu) x[u] = fmaxf(x[u], 0.f);
                if (nb + u >= p.N) x[u] = 0.f;
            }
            if (EPI == EPI_PLANES) {
                unsigned short hb[8], lb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { hb[u] = bf16_bits(x[u]); lb[u] = bf16_bits(x[u] - bf16_to_f32(hb[u])); }
                uint4 ph, pl;
                ph.x = hb[0] | ((unsigned)hb[1] << 16); ph.y = hb[2] | ((unsigned)hb[3] << 16); ph.z = hb[4] | ((unsigned)hb[5] << 16); ph.w = hb[6] | ((unsigned)hb[7] << 16);
                pl.x = lb[0] | ((unsigned)lb[1] << 16); pl.y = lb[2] | ((unsigned)lb[3] << 16); pl.z = lb[4] | ((unsigned)lb[5] << 16); pl.w = lb[6] | ((unsigned)lb[7] << 16);
                const int64_t o = boff + (int64_t)m * p.ldp + nb;
                *reinterpret_cast<uint4*>(p.Ph + o) = ph;
                *reinterpret_cast<uint4*>(p.Pl + o) = pl;
            } else {
                float* dst = p.C + boff + (int64_t)m * p.ldc_m + (int64_t)nb * p.ldc_n;
                if (vecC && nb + 8 <= p.N) {
                    reinterpret_cast<float4*>(dst)[0] = make_float4(x[0], x[1], x[2], x[3]);
                    reinterpret_cast<float4*>(dst)[1] = make_float4(x[4], x[5], x[6], x[7]);
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (nb + u < p.N) dst[(int64_t)u * p.ldc_n] = x[u];
                }
            }
        }
        return;
    }
    float* C = p.C + b1 * p.sC1 + b2 * p.sC2;
    if (EPI == EPI_INTERLEAVE2) {
        // GEMM rows (2m, 2m+1) are the two glimpses of one (v,q) row: registers e and e+1 (e even) of a lane are
        // adjacent floats in out[b, vq, a, 0:2] -> one 8-B store per lane, 256 contiguous bytes per half-wave
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= p.N) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);   // even
                    if (m < p.M) {
                        float2 v2 = make_float2(acc[i][j][e], acc[i][j][e + 1]);
                        *reinterpret_cast<float2*>(C + (int64_t)(m >> 1) * p.ldc_m + (int64_t)n * 2) = v2;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= p.N) continue;
        const float sc = p.scale ? p.scale[n / p.scale_div] : 1.f;
        const float bi = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float x = acc[i][j][e] * sc + bi;
                    if (p.relu) x = fmaxf(x, 0.f);
                    if (EPI == EPI_INTERLEAVE) C[(int64_t)(m / p.gdiv) * p.ldc_m + (m % p.gdiv) + (int64_t)n * p.ldc_n] = x;
                    else                       C[(int64_t)m * p.ldc_m + (int64_t)n * p.ldc_n] = x;
                }
            }
        }
    }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

// A "planes" buffer: [hi | lo], each rows_alloc x Kp bf16, lo = hi + rows_alloc*Kp.  rows_alloc includes slack rows so
// that the last tile's DMA of rows >= `rows` stays inside the allocation (their values only reach discarded outputs).
int planes_kp(int K) { return round_up(K, BK); }
size_t planes_bytes(int64_t rows_alloc, int K) { return 2 * sizeof(unsigned short) * (size_t)rows_alloc * planes_kp(K); }

int split_planes(const float* x, int64_t ld, int64_t rows, int K, unsigned short* hi, unsigned short* lo, hipStream_t st) {
    const int Kp = planes_kp(K);
    const int64_t n = rows * (Kp >> 3);
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ld, rows, K, Kp, hi, lo);
    return launch_status("split_planes");
}

int gemm_nt_planes(const PlaneGemmArgs& a, hipStream_t st) {
    PlaneGemmP p{};
    p.Ah = a.Ah; p.Al = a.Al; p.Bh = a.Bh; p.Bl = a.Bl; p.C = a.C;
    p.lda = a.lda; p.ldb = a.ldb; p.ldc_m = a.ldc_m; p.ldc_n = a.ldc_n;
    p.sA1 = a.sA1; p.sA2 = a.sA2; p.sB1 = a.sB1; p.sB2 = a.sB2; p.sC1 = a.sC1; p.sC2 = a.sC2;
    p.nb2 = a.nb2; p.M = a.M; p.N = a.N; p.Kp = a.Kp;
    p.scale = a.scale; p.scale_div = a.scale_div > 0 ? a.scale_div : 1; p.bias = a.bias; p.relu = a.relu;
    p.Ph = a.Ph; p.Pl = a.Pl; p.ldp = a.ldp; p.Np = a.Np; p.gdiv = a.gdiv > 0 ? a.gdiv : 1;
    if (a.Kp % BK != 0 || (a.lda & 7) || (a.ldb & 7)) return fail(CTI_E_ALIGN, "gemm_nt_planes: Kp=%d lda=%lld ldb=%lld", a.Kp, (long long)a.lda, (long long)a.ldb);
    const int ncols = a.epi == 1 ? a.Np : a.N;
    const long long nb = (long long)a.nb1 * a.nb2;
    const int tiles_n = (ncols + BN - 1) / BN;
    // 256-row tiles (8 waves, 144 KiB ring) once they still give every CU work; else 128-row tiles (4 waves, 96 KiB)
    const long long t256 = nb * ((a.M + 255) / 256) * tiles_n;
    const int wm = (a.M > 128 && t256 >= 256) ? 4 : 2;
    const int BMt = wm * 64;
    const long long total = nb * ((a.M + BMt - 1) / BMt) * tiles_n;
    if (total > 0x7fffffffLL) return fail(CTI_E_SHAPE, "gemm_nt_planes: %lld tiles exceed the grid", total);
    dim3 grid((unsigned)total, 1, 1);
    const size_t lds = (size_t)NSTAGE * 2 * (BMt * ROW_BYTES + B_PLANE);
    const int epi = (a.epi == 3 && p.gdiv == 2 && a.ldc_n == 2) ? 2 : a.epi;
    const int key = (wm == 4 ? 8 : 0) + (a.terms == 3 ? 4 : 0) + epi;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
#define CTI_ATTR(T, E, W) { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_kernel<T, E, W>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                            if (e_ != hipSuccess) return fail((int)e_, "gemm_nt_planes: hipFuncSetAttribute: %s", hipGetErrorString(e_)); }
    if (attr_dev != dev) {
        CTI_ATTR(1, 0, 2) CTI_ATTR(1, 1, 2) CTI_ATTR(1, 2, 2) CTI_ATTR(1, 3, 2) CTI_ATTR(3, 0, 2) CTI_ATTR(3, 1, 2) CTI_ATTR(3, 2, 2) CTI_ATTR(3, 3, 2)
        CTI_ATTR(1, 0, 4) CTI_ATTR(1, 1, 4) CTI_ATTR(1, 2, 4) CTI_ATTR(1, 3, 4) CTI_ATTR(3, 0, 4) CTI_ATTR(3, 1, 4) CTI_ATTR(3, 2, 4) CTI_ATTR(3, 3, 4)
        attr_dev = dev;
    }
#undef CTI_ATTR
    switch (key) {
#define CTI_L(T, E, W) hipLaunchKernelGGL((gemm_planes_kernel<T, E, W>), grid, dim3(W * 128), lds, st, p); break;
        case 0: CTI_L(1, 0, 2) case 1: CTI_L(1, 1, 2) case 2: CTI_L(1, 2, 2) case 3: CTI_L(1, 3, 2)
        case 4: CTI_L(3, 0, 2) case 5: CTI_L(3, 1, 2) case 6: CTI_L(3, 2, 2) case 7: CTI_L(3, 3, 2)
        case 8: CTI_L(1, 0, 4) case 9: CTI_L(1, 1, 4) case 10: CTI_L(1, 2, 4) case 11: CTI_L(1, 3, 4)
        case 12: CTI_L(3, 0, 4) case 13: CTI_L(3, 1, 4) case 14: CTI_L(3, 2, 4) case 15: CTI_L(3, 3, 4)
#undef CTI_L
        default: return fail(CTI_E_UNSUPPORTED, "gemm_nt_planes: epi=%d terms=%d", a.epi, a.terms);
    }
    return launch_status("gemm_nt_planes");
}

}  // namespace cti
