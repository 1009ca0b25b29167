// cti_common.h -- shared host-side helpers of libcti_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/cti_hip.h"

namespace cti {

// thread-local last error text (the only mutable state of the library)
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// after a kernel launch: pick up launch errors without synchronising
inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return CTI_OK;
}

#define CTI_REQUIRE_PTR(p)  do { if ((p) == nullptr) return ::cti::fail(CTI_E_NULL, "%s: %s is NULL", __func__, #p); } while (0)
#define CTI_REQUIRE(cond, code, ...) do { if (!(cond)) return ::cti::fail((code), __VA_ARGS__); } while (0)

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

// ---- strided, batched NT GEMM (both operands K-contiguous) used by the projection layers and the PARALIND core
struct GemmP {
    const float* A; const float* B; float* C;
    int64_t lda, ldb, ldc_m, ldc_n;            // element strides
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;      // batch strides: z = b1 * nb2 + b2
    int nb1, nb2;
    int M, N, K;
    const float* scale; int scale_div;         // epilogue: acc * scale[n / scale_div] (NULL = 1)
    const float* bias;                         // + bias[n] (NULL = 0)
    int relu;
};
int gemm_nt_f32(const GemmP& p, hipStream_t st);

}  // namespace cti
