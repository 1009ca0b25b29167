// cti_gemm_f32.hip -- exact-fp32 NT GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[z][m, n] = act( scale[n / scale_div] * sum_k A[z][m, k] * B[z][n, k] + bias[n] )
//
// Both operands are K-contiguous (activations (rows, in) and nn.Linear weights (out, in)), which is the only
// GEMM shape the CTI hot path has: FCNet layers (reference src/fc.py:33-34), the packed rank nets
// (src/tc.py:47-49) and the mode-3 product + rank sum of the PARALIND core (src/Tensor.py:16-20, src/tc.py:50).
//
// Tile: 128 x 128 x 32 per 256-thread workgroup (4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles of 32 x 32).
// Operand tiles are staged global -> registers -> LDS TRANSPOSED ([k][row], row stride 129 floats) so that the
// MFMA fragment read (lane l: row l&31, k = 2*kk + (l>>5)) is one conflict-free ds_read_b32 per operand tile and
// the transposing ds_write_b32 (lane: 4 consecutive k of one row) is conflict-free as well (4*129 mod 32 = 4).
// Register prefetch of tile t+1 overlaps the 64 MFMAs (64 cycles each) a wave issues per tile; two LDS buffers,
// one barrier per K-step.  The result of this instruction is bit-for-bit a k-ordered fmaf chain, so the path is
// exact fp32 (no split, no reduced precision).
#include "cti_common.h"

namespace cti {

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = 129, NT = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Tile4 { float4 v[4]; };

// one thread's share of a 128 x 32 operand tile: 4 passes of (row = t>>3 + 32*p, k = 4*(t&7) .. +3)
__device__ __forceinline__ void load_tile(Tile4& r, const float* __restrict__ base, int64_t ld, int rows_left,
                                          int k0, int K, bool vec, int t) {
    const int kc = (t & 7) * 4;
    const int k = k0 + kc;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (t >> 3) + 32 * p;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows_left) {
            const float* src = base + (int64_t)row * ld + k;
            if (vec && k + 4 <= K) {
                x = *reinterpret_cast<const float4*>(src);
            } else {
                if (k + 0 < K) x.x = src[0];
                if (k + 1 < K) x.y = src[1];
                if (k + 2 < K) x.z = src[2];
                if (k + 3 < K) x.w = src[3];
            }
        }
        r.v[p] = x;
    }
}

__device__ __forceinline__ void store_tile(const Tile4& r, float* lds /* [BK][LDT] */, int t) {
    const int kc = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (t >> 3) + 32 * p;
        lds[(kc + 0) * LDT + row] = r.v[p].x;
        lds[(kc + 1) * LDT + row] = r.v[p].y;
        lds[(kc + 2) * LDT + row] = r.v[p].z;
        lds[(kc + 3) * LDT + row] = r.v[p].w;
    }
}

__global__ __launch_bounds__(NT) void gemm_nt_f32_kernel(GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BK][LDT]
    float* Bs = smem + 2 * BK * LDT;        // [2][BK][LDT]

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    int z, tm, tn;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, z, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int b1 = z / p.nb2, b2 = z % p.nb2;

    const float* A = p.A + b1 * p.sA1 + b2 * p.sA2 + (int64_t)m0 * p.lda;
    const float* B = p.B + b1 * p.sB1 + b2 * p.sB2 + (int64_t)n0 * p.ldb;
    const bool vecA = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool vecB = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    const int rowsA = p.M - m0, rowsB = p.N - n0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    Tile4 ra, rb;
    load_tile(ra, A, p.lda, rowsA, 0, p.K, vecA, t);
    load_tile(rb, B, p.ldb, rowsB, 0, p.K, vecB, t);
    store_tile(ra, As, t);
    store_tile(rb, Bs, t);
    __syncthreads();

    const int frag_row = lane & 31, frag_k = lane >> 5;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            load_tile(ra, A, p.lda, rowsA, (kt + 1) * BK, p.K, vecA, t);
            load_tile(rb, B, p.ldb, rowsB, (kt + 1) * BK, p.K, vecB, t);
        }
        const float* as = As + cur * BK * LDT + wm * 64 + frag_row;
        const float* bs = Bs + cur * BK * LDT + wn * 64 + frag_row;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + frag_k;
            const float a0 = as[k * LDT], a1 = as[k * LDT + 32];
            const float b0 = bs[k * LDT], b1v = bs[k * LDT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1v, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1v, acc[1][1], 0, 0, 0);
        }
        if (more) {
            store_tile(ra, As + (cur ^ 1) * BK * LDT, t);
            store_tile(rb, Bs + (cur ^ 1) * BK * LDT, t);
        }
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: D[i][j]: j = lane & 31, i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* C = p.C + b1 * p.sC1 + b2 * p.sC2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= p.N) continue;
        const float sc = p.scale ? p.scale[b1 * p.scale_bs + n / p.scale_div] : 1.f;
        const float bi = p.bias ? p.bias[b1 * p.bias_bs + n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float x = acc[i][j][e] * sc + bi;
                    if (p.relu) x = relu_nan(x);
                    C[(int64_t)m * p.ldc_m + (int64_t)n * p.ldc_n] = x;
                }
            }
        }
    }
}

}  // namespace

int gemm_nt_f32(const GemmP& p, hipStream_t st) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const long long nb = (long long)p.nb1 * p.nb2;
    const long long total = nb * tiles_m * tiles_n;
    if (total > 0x7fffffffLL) return fail(CTI_E_SHAPE, "gemm_nt_f32: %lld tiles exceed the grid", total);
    dim3 grid((unsigned)total, 1, 1);
    const size_t lds = sizeof(float) * 4 * BK * LDT;          // 66,048 B: above the 64 KiB default, so opt in once
    static thread_local int attr_dev = -1;
    int dev = 0;
    hipGetDevice(&dev);
    if (dev != attr_dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f32_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "gemm_nt_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    hipLaunchKernelGGL(gemm_nt_f32_kernel, grid, dim3(NT), lds, st, p);
    return launch_status("gemm_nt_f32");
}

}  // namespace cti
