// cti_optim.hip -- the update half of the data-parallel training step (reference src/FFOE/trainer.py:221-269,
// src/utils.py:323-328, torch.optim.Adamax of src/FFOE/train.py:34) on ONE flat fp32 buffer per quantity:
//   kernel 0: the gradients autograd left in separate tensors -> their 256-B aligned slots of the flat buffer, everything else zeroed
//             (ONE launch instead of one accumulate-add per parameter: the FFOE CTI model has 344 parameters, 288 of them rank nets)
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

// One thread per float4 of [first, last); the 16 lanes of a 64-float granule (the parameters' alignment unit) find their entry by the same
// binary search.  The entries travel in the kernel arguments (no table upload to race with the next step): up to GATHER_ENTRIES per launch.
constexpr int GATHER_ENTRIES = 160;
struct GatherBatch {
    const float* src[GATHER_ENTRIES];
    int64_t off[GATHER_ENTRIES];
    int64_t cnt[GATHER_ENTRIES];
    int n;
};

__global__ __launch_bounds__(256) void flat_gather_kernel(const GatherBatch tb, float* flat, int64_t first, int64_t last) {
    const int64_t e = first + ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= last) return;
    int lo = 0, hi = tb.n;                                        // last entry whose slot starts at or before e
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tb.off[mid] <= e) lo = mid; else hi = mid;
    }
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tb.n > 0) {
        const int64_t off = tb.off[lo], cnt = tb.cnt[lo];
        const float* src = tb.src[lo];
        const int64_t rel = e - off;
        if (rel >= 0 && rel < cnt && src) {
            if (src == flat + off && rel + 4 <= cnt) return;      // this gradient already lives in its slot (only its last float4 is re-padded)
            if (rel + 4 <= cnt && ((reinterpret_cast<uintptr_t>(src + rel) & 15) == 0)) {
                out = *reinterpret_cast<const float4*>(src + rel);
            } else {
                out.x = src[rel];
                if (rel + 1 < cnt) out.y = src[rel + 1];
                if (rel + 2 < cnt) out.z = src[rel + 2];
                if (rel + 3 < cnt) out.w = src[rel + 3];
            }
        }
    }
    *reinterpret_cast<float4*>(flat + e) = out;
}

__global__ __launch_bounds__(256) void adamax_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ u,
                                                     int64_t n, const float* __restrict__ partial, int nparts, float max_norm, float lr_t,
                                                     float b1, float b2, float eps, float* __restrict__ norm_out,
                                                     const float* __restrict__ lr_dev, const long long* __restrict__ step_dev) {
    __shared__ float red[4];
    __shared__ float coef_s;
    // graph-safe form: learning rate and the count of completed steps live in device memory (a captured hipGraph replays with fresh values)
    if (step_dev) lr_t = lr_dev[0] / (1.f - powf(b1, (float)(step_dev[0] + 1)));
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

extern "C" int cti_flat_gather(const int64_t* table /* HOST */, int n_entries, float* flat, int64_t n, void* stream) {
    CTI_REQUIRE_PTR(flat);
    CTI_REQUIRE(n > 0 && n % 4 == 0 && n_entries >= 0 && (n_entries == 0 || table) && (reinterpret_cast<uintptr_t>(flat) & 15) == 0, CTI_E_SHAPE,
                "cti_flat_gather: n=%lld (a multiple of 4, flat 16-B aligned) entries=%d", (long long)n, n_entries);
    int64_t prev_end = 0;
    for (int i = 0; i < n_entries; ++i) {
        const int64_t off = table[3 * i + 1], cnt = table[3 * i + 2];
        CTI_REQUIRE(off % 4 == 0 && off >= prev_end && cnt >= 0 && off + cnt <= n, CTI_E_SHAPE,
                    "cti_flat_gather: entry %d (slot %lld, count %lld) overlaps its predecessor, is unaligned or leaves the buffer", i, (long long)off, (long long)cnt);
        prev_end = off + cnt;
    }
    int i0 = 0;
    do {                                                          // at least one launch: with no entries the whole buffer is zeroed
        GatherBatch tb;
        tb.n = n_entries - i0 < GATHER_ENTRIES ? n_entries - i0 : GATHER_ENTRIES;
        for (int j = 0; j < tb.n; ++j) {
            tb.src[j] = reinterpret_cast<const float*>(static_cast<uintptr_t>(table[3 * (i0 + j)]));
            tb.off[j] = table[3 * (i0 + j) + 1];
            tb.cnt[j] = table[3 * (i0 + j) + 2];
        }
        const int64_t first = i0 == 0 ? 0 : tb.off[0];
        const int64_t last = i0 + tb.n >= n_entries ? n : table[3 * (i0 + tb.n) + 1];
        if (last > first) {
            const int64_t nb = ((last - first) / 4 + 255) / 256;
            hipLaunchKernelGGL(flat_gather_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), tb, flat, first, last);
        }
        i0 += tb.n;
    } while (i0 < n_entries);
    return launch_status("cti_flat_gather");
}

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
                       partial, OPT_BLOCKS, max_norm, lr_t, beta1, beta2, eps, grad_norm_out, nullptr, nullptr);
    return launch_status("cti_adamax_step");
}

extern "C" int cti_adamax_step_g(float* param, const float* grad, float* exp_avg, float* exp_inf, int64_t n, const float* partial, float max_norm,
                                 const float* lr_dev, float beta1, float beta2, float eps, const int64_t* steps_done_dev, float* grad_norm_out, void* stream) {
    CTI_REQUIRE_PTR(param); CTI_REQUIRE_PTR(grad); CTI_REQUIRE_PTR(exp_avg); CTI_REQUIRE_PTR(exp_inf); CTI_REQUIRE_PTR(partial);
    CTI_REQUIRE_PTR(lr_dev); CTI_REQUIRE_PTR(steps_done_dev);
    CTI_REQUIRE(n > 0, CTI_E_SHAPE, "cti_adamax_step_g: n=%lld", (long long)n);
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(adamax_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_inf, n,
                       partial, OPT_BLOCKS, max_norm, 0.f, beta1, beta2, eps, grad_norm_out, lr_dev, reinterpret_cast<const long long*>(steps_done_dev));
    return launch_status("cti_adamax_step_g");
}
