// Do VALU instructions hide behind MFMAs on a gfx950 SIMD?  512-thread workgroups (two waves per SIMD, one workgroup per CU), every wave
// loops over NM independent v_mfma_f32_32x32x16_f16 (6 accumulators, as in gemm_f16f6_kernel) and NV independent VALU ops (v_fma_f32 on
// private registers; optionally v_cvt_scalef32_pk32_fp6_f16).  If the two kinds overlapped, time(NM, NV) ~ max(NM t_m, NV t_v); if they share
// an issue port, ~ NM t_m + NV t_v.  Prints cycles per loop iteration per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o mb_issue mb_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NM, int NV, int NC>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* cyc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = src[(t + i + e) & 1023];
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)src[(t + e) & 1023]; b[e] = (_Float16)src[(t + 8 + e) & 1023]; }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = src[(t * 3 + e) & 1023];
    f16x32 cin;
    for (int e = 0; e < 32; ++e) cin[e] = (_Float16)src[(t + e) & 1023];
    unsigned csum = 0;
    const float m1 = src[5], m2 = src[6];
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < (NM > NV ? NM : (NV > NC ? NV : NC)); ++u) {
            if (u < NM) acc[u % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % 6], 0, 0, 0);
            if (u < NV) v[u % 8] = __builtin_fmaf(v[u % 8], m1, m2);
            if (u < NC) { u32x6 o; asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(o) : "v"(cin), "v"(m1)); csum += o[0]; }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[t] = s + (float)csum;
    if (t == 0) *cyc = c1 - c0;
}

template <int NM, int NV, int NC>
static void run(const float* src, float* out, unsigned long long* cyc, const char* what) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NM, NV, NC>), dim3(256), dim3(512), 0, 0, src, out, iters, cyc);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<NM, NV, NC>), dim3(256), dim3(512), 0, 0, src, out, iters, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-46s NM=%2d NV=%3d NC=%d: %8.1f shader cycles per iteration (wave 0), %7.3f us per 1000 iterations\n", what, NM, NV, NC, (double)c / iters, ms * 1e3 / iters * 1000);
}

int main() {
    float* src; float* out; unsigned long long* cyc;
    CK(hipMalloc(&src, 4096)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() % 1000) / 1000.f - 0.5f;
    CK(hipMemcpy(src, h, 4096, hipMemcpyHostToDevice));
    run<18, 0, 0>(src, out, cyc, "18 MFMA");
    run<0, 100, 0>(src, out, cyc, "100 VALU (v_fma_f32)");
    run<18, 100, 0>(src, out, cyc, "18 MFMA + 100 VALU interleaved");
    run<18, 50, 0>(src, out, cyc, "18 MFMA + 50 VALU");
    run<18, 18, 0>(src, out, cyc, "18 MFMA + 18 VALU");
    run<0, 0, 5>(src, out, cyc, "5 cvt_pk32_fp6_f16");
    run<18, 0, 5>(src, out, cyc, "18 MFMA + 5 cvt");
    run<18, 100, 5>(src, out, cyc, "18 MFMA + 100 VALU + 5 cvt");
    return 0;
}
