// Micro-benchmark + layout check for the f16 + block-scaled-fp6 split product (gfx950).
//   part 1: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with e2m3 (fp6) operands, checked with exactly representable data
//   part 2: sustained MFMA throughput on random operands in registers: 12 x bf16 32x32x16 per K=64 (bf16x3) vs 4 x f16 32x32x16 + 2 x fp6 32x32x64
// build: hipcc -O3 --offload-arch=gfx950 -o mb_f16f6 mb_f16f6.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- part 1 ---------------------------------------------------------------------------------------------------------
// a, b: [32 rows][64 k] fp6 codes (one byte each, low 6 bits); sa, sb: [32][2] e8m0 scale bytes per (row, 32-block)
__global__ void layout_kernel(const uint8_t* a, const uint8_t* b, const uint8_t* sa, const uint8_t* sb, float* out) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    uint32_t av[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 32; ++j) {
        const uint64_t ca = a[r * 64 + h * 32 + j] & 63, cb = b[r * 64 + h * 32 + j] & 63;
        const int bit = 6 * j, w = bit >> 5, s = bit & 31;
        av[w] |= (uint32_t)(ca << s); if (s > 26) av[w + 1] |= (uint32_t)(ca >> (32 - s));
        bv[w] |= (uint32_t)(cb << s); if (s > 26) bv[w + 1] |= (uint32_t)(cb >> (32 - s));
    }
    i32x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (int)av[i]; B[i] = (int)bv[i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    const int scA = sa[r * 2 + h], scB = sb[r * 2 + h];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 2 /*A fp6 e2m3*/, 2 /*B fp6 e2m3*/, 0, scA, 0, scB);
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        out[row * 32 + r] = c[e];
    }
}

static float e2m3_val(int code) {
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    float v = e == 0 ? m / 8.f : (1.f + m / 8.f) * (float)(1 << (e - 1));
    return s ? -v : v;
}

// ---- part 2 ---------------------------------------------------------------------------------------------------------
template <int MODE, int TM, int TN>
__global__ __launch_bounds__(512) void rate_kernel(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // random operand registers (distinct per lane)
    uint4 ra[TM][4], rb[TN][4];
    for (int i = 0; i < TM; ++i) for (int u = 0; u < 4; ++u) ra[i][u] = src[(t * 8 + i * 4 + u) & 0xffff];
    for (int j = 0; j < TN; ++j) for (int u = 0; u < 4; ++u) rb[j][u] = src[(t * 8 + 64 + j * 4 + u) & 0xffff];
    for (int it = 0; it < iters; ++it) {
        // one K = 64 step of a (TM*32) x (TN*32) wave tile
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (MODE == 0) {                 // bf16x3: 4 k-steps x 3 products
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, ra[i][u]), al = __builtin_bit_cast(bf16x8, ra[i][(u + 1) & 3]);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, rb[j][u]), bl = __builtin_bit_cast(bf16x8, rb[j][(u + 2) & 3]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f16x8 ah = __builtin_bit_cast(f16x8, ra[i][u]);
                        const f16x8 bh = __builtin_bit_cast(f16x8, rb[j][u]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                    }
                    if (MODE >= 2) {
                        i32x8 A6, B6, A6l, B6l;
                        A6[0] = ra[i][0].x; A6[1] = ra[i][0].y; A6[2] = ra[i][0].z; A6[3] = ra[i][0].w; A6[4] = ra[i][1].x; A6[5] = ra[i][1].y; A6[6] = ra[i][1].z; A6[7] = ra[i][1].w;
                        B6[0] = rb[j][0].x; B6[1] = rb[j][0].y; B6[2] = rb[j][0].z; B6[3] = rb[j][0].w; B6[4] = rb[j][1].x; B6[5] = rb[j][1].y; B6[6] = rb[j][1].z; B6[7] = rb[j][1].w;
                        A6l[0] = ra[i][2].x; A6l[1] = ra[i][2].y; A6l[2] = ra[i][2].z; A6l[3] = ra[i][2].w; A6l[4] = ra[i][3].x; A6l[5] = ra[i][3].y; A6l[6] = ra[i][3].z; A6l[7] = ra[i][3].w;
                        B6l[0] = rb[j][2].x; B6l[1] = rb[j][2].y; B6l[2] = rb[j][2].z; B6l[3] = rb[j][2].w; B6l[4] = rb[j][3].x; B6l[5] = rb[j][3].y; B6l[6] = rb[j][3].z; B6l[7] = rb[j][3].w;
                        if (MODE == 2) {        // fp6 corrections
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A6, B6l, acc[i][j], 2, 2, 0, 127, 0, 116);
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A6l, B6, acc[i][j], 2, 2, 0, 116, 0, 127);
                        } else {                // fp8 corrections
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A6, B6l, acc[i][j], 0, 0, 0, 127, 0, 116);
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A6l, B6, acc[i][j], 0, 0, 0, 116, 0, 127);
                        }
                    }
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[t] = s;
}

template <int MODE>
static void run_rate(const char* name, const uint4* src, float* out, int threads, int iters, double units_per_k64) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((rate_kernel<MODE, 2, 4>), dim3(256), dim3(threads), 0, 0, src, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    // steady state: 2 s of back-to-back launches, then time 5
    float ms = 0.f;
    for (int rep = 0; rep < 40; ++rep) hipLaunchKernelGGL((rate_kernel<MODE, 2, 4>), dim3(256), dim3(threads), 0, 0, src, out, iters);
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((rate_kernel<MODE, 2, 4>), dim3(256), dim3(threads), 0, 0, src, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double waves = 256.0 * threads / 64;
    const double alg_flops = waves * iters * 8.0 * (2.0 * 32 * 32 * 64);      // algorithmic products of a 64x128 wave tile per K=64
    printf("%-28s threads/WG %3d  %.3f ms  algorithmic %.1f TFLOP/s  issued(16-bit equiv) %.1f TFLOP/s\n", name, threads, ms, alg_flops / ms / 1e9,
           alg_flops * units_per_k64 / ms / 1e9);
}

int main() {
    // ---- part 1
    std::vector<uint8_t> a(32 * 64), b(32 * 64), sa(64), sb(64);
    srand(1);
    for (auto& x : a) x = rand() & 63;
    for (auto& x : b) x = rand() & 63;
    for (auto& x : sa) x = 127 + (rand() % 7) - 3;
    for (auto& x : sb) x = 127 + (rand() % 5) - 2;
    uint8_t *da, *db, *dsa, *dsb; float* dout;
    CK(hipMalloc(&da, a.size())); CK(hipMalloc(&db, b.size())); CK(hipMalloc(&dsa, 64)); CK(hipMalloc(&dsb, 64)); CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
    std::vector<float> out(1024);
    CK(hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < 64; ++k)
            ref += (double)e2m3_val(a[i * 64 + k]) * ldexp(1.0, sa[i * 2 + k / 32] - 127) * (double)e2m3_val(b[j * 64 + k]) * ldexp(1.0, sb[j * 2 + k / 32] - 127);
        maxerr = fmax(maxerr, fabs(ref - out[i * 32 + j])); maxref = fmax(maxref, fabs(ref));
    }
    printf("fp6 layout check: max |err| %.3g (max |ref| %.3g) -> %s\n", maxerr, maxref, maxerr <= 1e-5 * maxref ? "LAYOUT OK" : "LAYOUT MISMATCH");
    // ---- part 2
    const size_t nsrc = 65536;
    std::vector<uint32_t> rnd(nsrc * 4);
    for (auto& x : rnd) {
        // random f16 pairs in a sane range (exponent bits limited) so neither f16 nor bf16 / fp6 interpretations are all NaN / zero
        uint32_t lo = (rand() & 0x83ff) | (((rand() % 6) + 12) << 10), hi = (rand() & 0x83ff) | (((rand() % 6) + 12) << 10);
        x = lo | (hi << 16);
    }
    uint4* dsrc; float* dres;
    CK(hipMalloc(&dsrc, nsrc * 16)); CK(hipMalloc(&dres, 256 * 512 * 4));
    CK(hipMemcpy(dsrc, rnd.data(), nsrc * 16, hipMemcpyHostToDevice));
    const int iters = 4000;
    for (int threads = 256; threads <= 512; threads += 256) {
        run_rate<0>("bf16x3 (12 bf16 / K64)", dsrc, dres, threads, iters, 3.0);
        run_rate<1>("f16 only (4 f16 / K64)", dsrc, dres, threads, iters, 1.0);
        run_rate<2>("f16 + 2 fp6 (scaled K64)", dsrc, dres, threads, iters, 1.5);
        run_rate<3>("f16 + 2 fp8 (scaled K64)", dsrc, dres, threads, iters, 2.0);
    }
    return 0;
}
