// cti_optim.hip -- the update half of the data-parallel training step (reference src/FFOE/trainer.py:221-269,
// src/utils.py:323-328, torch.optim.Adamax of src/FFOE/train.py:34) on ONE flat fp32 buffer per quantity:
//   kernel 1: g *= 1/denom (denom = world_size * update_freq), per-workgroup partial sums of g^2          (HBM: read+write g)
//   kernel 2: norm = sqrt(sum partials); coef = min(1, max_norm / (norm + 1e-6)); g' = coef * g;
//             m = b1*m + (1-b1)*g';  u = max(b2*u, |g'| + eps);  p -= lr / (1 - b1^t) * m / u            (HBM: p, g, m, u)
// No host synchronisation: the clip coefficient never leaves the device (the reference syncs on grad_norm.item()).
#include "cti_common.h"

namespace cti {
namespace {

constexpr int OPT_BLOCKS = 1024;

__global__ __launch_bounds__(256) void scale_sumsq_kernel(float* __restrict__ g, int64_t n, float inv_denom, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(g + i) & 15) == 0)) {
            float4 x = *reinterpret_cast<float4*>(g + i);
            x.x *= inv_denom; x.y *= inv_denom; x.z *= inv_denom; x.w *= inv_denom;
            *reinterpret_cast<float4*>(g + i) = x;
            s = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, s))));
        } else {
            for (int64_t j = i; j < n && j < i + 4; ++j) { const float x = g[j] * inv_denom; g[j] = x; s = fmaf(x, x, s); }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamax_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ u,
                                                     int64_t n, const float* __restrict__ partial, int nparts, float max_norm, float lr_t,
                                                     float b1, float b2, float eps, float* __restrict__ norm_out) {
    __shared__ float red[4];
    __shared__ float coef_s;
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];      // every workgroup re-reduces the same partials in the same order
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        const float c = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
        coef_s = c < 1.f ? c : 1.f;
        if (blockIdx.x == 0 && norm_out) norm_out[0] = norm;
    }
    __syncthreads();
    const float coef = coef_s;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gg = g[i] * coef;
        const float mm = fmaf(b1, m[i], (1.f - b1) * gg);
        const float uu = fmaxf(b2 * u[i], fabsf(gg) + eps);
        m[i] = mm; u[i] = uu;
        p[i] -= lr_t * mm / uu;
    }
}

}  // namespace
}  // namespace cti

using namespace cti;

extern "C" size_t cti_optim_workspace_bytes(void) { return sizeof(float) * OPT_BLOCKS; }

extern "C" int cti_flat_scale_sumsq(float* grad, int64_t n, float inv_denom, float* partial /* cti_optim_workspace_bytes() */, void* stream) {
    CTI_REQUIRE_PTR(grad); CTI_REQUIRE_PTR(partial);
    CTI_REQUIRE(n > 0, CTI_E_SHAPE, "cti_flat_scale_sumsq: n=%lld", (long long)n);
    hipLaunchKernelGGL(scale_sumsq_kernel, dim3(OPT_BLOCKS), dim3(256), 0, as_stream(stream), grad, n, inv_denom, partial);
    return launch_status("cti_flat_scale_sumsq");
}

extern "C" int cti_adamax_step(float* param, const float* grad, float* exp_avg, float* exp_inf, int64_t n, const float* partial, float max_norm,
                               float lr, float beta1, float beta2, float eps, int step, float* grad_norm_out, void* stream) {
    CTI_REQUIRE_PTR(param); CTI_REQUIRE_PTR(grad); CTI_REQUIRE_PTR(exp_avg); CTI_REQUIRE_PTR(exp_inf); CTI_REQUIRE_PTR(partial);
    CTI_REQUIRE(n > 0 && step >= 1, CTI_E_SHAPE, "cti_adamax_step: n=%lld step=%d", (long long)n, step);
    const float lr_t = lr / (1.f - powf(beta1, (float)step));
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(adamax_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_inf, n,
                       partial, OPT_BLOCKS, max_norm, lr_t, beta1, beta2, eps, grad_norm_out);
    return launch_status("cti_adamax_step");
}
