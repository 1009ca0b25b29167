// cti_paralind.hip -- weight-norm scale, zero-row mask, T_eff scramble and the mode-1/mode-2 stage ("M build")
// of the PARALIND core.  All HBM/L2-bound helpers: coalesced loads along the innermost axis, wave-shuffle
// reductions (64-wide), no GEMM reshaping.
#include "cti_common.h"
#include "cti_f16f6.h"

namespace cti {
namespace {

// ---- scale[i] = g[i] / ||V_i||_F ------------------------------------------------------------------------------
// One 1024-thread workgroup per matrix; float4 loads when the slice is 16-B aligned; the block sum is a fixed
// tree (wave shuffles, then 16 partials through LDS) so the result is run-to-run deterministic.
__global__ __launch_bounds__(1024) void wn_scale_kernel(const float* __restrict__ wv, const float* __restrict__ g,
                                                        float* __restrict__ scale, int64_t elems) {
    __shared__ float part[16];
    const int i = blockIdx.x, t = threadIdx.x;
    const float* v = wv + (int64_t)i * elems;
    float s = 0.f;
    if (((reinterpret_cast<uintptr_t>(v) & 15) == 0) && ((elems & 3) == 0)) {
        const float4* v4 = reinterpret_cast<const float4*>(v);
        const int64_t n4 = elems >> 2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int64_t j = t; j < n4; j += 1024) {
            const float4 x = v4[j];
            s0 = fmaf(x.x, x.x, s0); s1 = fmaf(x.y, x.y, s1); s2 = fmaf(x.z, x.z, s2); s3 = fmaf(x.w, x.w, s3);
        }
        s = (s0 + s1) + (s2 + s3);
    } else {
        for (int64_t j = t; j < elems; j += 1024) s = fmaf(v[j], v[j], s);
    }
    s = wave_sum(s);
    if ((t & 63) == 0) part[t >> 6] = s;
    __syncthreads();
    if (t < 64) {
        float x = t < 16 ? part[t] : 0.f;
        x = wave_sum(x);
        if (t == 0) scale[i] = g[i] / sqrtf(x);
    }
}

// ---- the same for several layers in two launches (used by cti_tcnet_forward) -------------------------------------
// Launch 1: one 256-thread workgroup per 4,096-element chunk of any matrix -> partial sum of squares (fixed tree).
// Launch 2: one wave per matrix adds its chunk partials in index order -> scale.  Deterministic, and a 1M-element
// Tucker weight is read by 64 CUs instead of one.
__global__ __launch_bounds__(256) void wn_partial_kernel(WnBatch d, float* __restrict__ partial) {
    __shared__ float part[4];
    int c = blockIdx.x, e = 0;
    while (e + 1 < d.n && c >= d.chunk_begin[e + 1]) ++e;                 // which layer this chunk belongs to
    c -= d.chunk_begin[e];
    const int cpm = d.chunks_per_mat[e];
    const int mat = c / cpm, ch = c % cpm;
    const float* v = d.wv[e] + (int64_t)mat * d.elems[e];
    const int64_t lo = (int64_t)ch * WN_CHUNK, hi = min(d.elems[e], lo + WN_CHUNK);
    float s = 0.f;
    for (int64_t j = lo + threadIdx.x; j < hi; j += 256) s = fmaf(v[j], v[j], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(64) void wn_final_kernel(WnBatch d, const float* __restrict__ partial) {
    int m = blockIdx.x, e = 0;
    while (e + 1 < d.n && m >= d.mat_begin[e + 1]) ++e;
    m -= d.mat_begin[e];
    const int cpm = d.chunks_per_mat[e];
    const float* p = partial + d.chunk_begin[e] + m * cpm;
    float s = 0.f;
    for (int c = threadIdx.x; c < cpm; c += 64) s += p[c];
    s = wave_sum(s);
    if (threadIdx.x == 0) d.scale[e][m] = d.g[e][m] / sqrtf(s);
}

// ---- mask[r] = every element of row r is +-0 ---------------------------------------------------------------------
// One wave per row; OR of the magnitude bits, so -0.0 counts as zero and NaN / inf / subnormals do not.
// `v` as 32-bit words: a row is `dimw` words (fp32: one element each, magmask 0x7fffffff; bf16 (round 5): two elements each, magmask 0x7fff7fff).
__global__ __launch_bounds__(256) void zero_row_mask_kernel(const unsigned* __restrict__ v, int64_t ldw,
                                                            uint8_t* __restrict__ mask, int64_t rows, int dimw, unsigned magmask) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const unsigned* p = v + row * ldw;
    unsigned bits = 0u;
    if (((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ((dimw & 3) == 0)) {
        const uint4* p4 = reinterpret_cast<const uint4*>(p);
        for (int j = lane; j < (dimw >> 2); j += 64) {
            const uint4 x = p4[j];
            bits |= (x.x | x.y | x.z | x.w);
        }
    } else {
        for (int j = lane; j < dimw; j += 64) bits |= p[j];
    }
    const bool nz = (bits & magmask) != 0u;
    const bool any_nz = __any(nz);
    if (lane == 0) mask[row] = any_nz ? 0 : 1;
}

// ---- T_eff scramble -------------------------------------------------------------------------------------------
// For fixed (r, i) the reference's mode-1 step (src/Tensor.py:6-8) transposes the (j,k) axes of T[i] = (J,K,G),
// flattens in (k,j,g) order and re-reads the flat axis as (G,K,J), exposed as (j,k,g) by .transpose(4,2):
//   T_eff[r,i,j,k,g] = flat[(g*K + k)*J + j],  flat[(k'*J + j')*G + g'] = T[r,i,j',k',g'].
__device__ __forceinline__ int teff_src_index(int j, int k, int g, int J, int K, int G) {
    const int f = (g * K + k) * J + j;
    const int gs = f % G, js = (f / G) % J, ks = f / (G * J);
    return (js * K + ks) * G + gs;
}
__global__ __launch_bounds__(256) void teff_kernel(const float* __restrict__ src, float* __restrict__ dst, int J, int K, int G,
                                                   int64_t total, int inverse) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int inner = J * K * G;
    const int64_t ri = idx / inner;
    const int e = (int)(idx - ri * inner);
    const int g = e % G, k = (e / G) % K, j = e / (G * K);
    const int s = teff_src_index(j, k, g, J, K, G);
    if (!inverse) dst[idx] = src[ri * inner + s];
    else          dst[ri * inner + s] = src[idx];
}

// ---- M build: modes 1 and 2 of the core ------------------------------------------------------------------------
// One workgroup per (b, v).  Per rank r: X[j,k,g] = sum_i T_eff[r,i,j,k,g] * Vr[b,v,r,i] (T_eff read coalesced along
// its contiguous (j,k,g) axis, L2-resident: R*hr^3*G floats = 1 MiB at the default sizes), kept in LDS; then
// M[b,v,q,g,r*K+k] = sum_j X[j,k,g] * Qr[b,q,r,j].  35 MFLOP per sample at C2: three orders of magnitude below
// the mode-3 GEMM that consumes M, so this stage is written for simplicity, not for the MFMA.
__global__ __launch_bounds__(256) void mbuild_kernel(const float* __restrict__ Vr, const float* __restrict__ Qr,
                                                     const float* __restrict__ Teff, float* __restrict__ M,
                                                     int V, int Q, int R, int I, int J, int K, int G) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int inner = J * K * G;             // (j,k,g)
    float* X = sm;                           // [inner]
    float* vs = sm + inner;                  // [I]   Vr slice
    float* qs = vs + I;                      // [Q*J] Qr slice
    const int bv = blockIdx.x;               // b*V + v
    const int b = bv / V;
    const int t = threadIdx.x;
    const int KK = R * K;
    const float* vrow = Vr + (int64_t)bv * R * I;
    const float* qbase = Qr + (int64_t)b * Q * R * J;
    float* mbase = M + (int64_t)bv * Q * G * KK;
    for (int r = 0; r < R; ++r) {
        for (int e = t; e < I; e += 256) vs[e] = vrow[r * I + e];
        for (int e = t; e < Q * J; e += 256) qs[e] = qbase[(int64_t)(e / J) * R * J + r * J + (e % J)];
        __syncthreads();
        const float* Tr = Teff + (int64_t)r * I * inner;
        for (int e = t; e < inner; e += 256) {
            float s = 0.f;
            for (int i = 0; i < I; ++i) s = fmaf(Tr[(int64_t)i * inner + e], vs[i], s);
            X[e] = s;
        }
        __syncthreads();
        // outputs (q, g, k): k fastest so that the K floats of one (q,g) row segment are written together
        for (int o = t; o < Q * G * K; o += 256) {
            const int k = o % K, g = (o / K) % G, q = o / (K * G);
            float s = 0.f;
            for (int j = 0; j < J; ++j) s = fmaf(X[(j * K + k) * G + g], qs[q * J + j], s);
            mbase[((int64_t)q * G + g) * KK + r * K + k] = s;
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace cti

using namespace cti;

namespace cti {
int mbuild_fast(const float* Vr, const float* Qr, const float* Teff, float* Mf, unsigned short* Mh, unsigned short* Ml, int B,
                int V, int Q, int R, int hr, int G, int64_t ldm_or_pitch, hipStream_t st);
size_t wn_batch_partials(const WnBatch& d) { return (size_t)d.chunk_begin[d.n]; }
void wn_batch_finish(WnBatch& d) {
    int cb = 0, mb = 0;
    for (int e = 0; e < d.n; ++e) {
        d.chunks_per_mat[e] = (int)((d.elems[e] + WN_CHUNK - 1) / WN_CHUNK);
        d.chunk_begin[e] = cb; d.mat_begin[e] = mb;
        cb += d.chunks_per_mat[e] * d.n_mats[e]; mb += d.n_mats[e];
    }
    d.chunk_begin[d.n] = cb; d.mat_begin[d.n] = mb;
}
int wn_scale_batch(const WnBatch& d, float* partial, hipStream_t st) {
    hipLaunchKernelGGL(wn_partial_kernel, dim3((unsigned)d.chunk_begin[d.n]), dim3(256), 0, st, d, partial);
    int rc = launch_status("wn_scale_batch/partial"); if (rc) return rc;
    hipLaunchKernelGGL(wn_final_kernel, dim3((unsigned)d.mat_begin[d.n]), dim3(64), 0, st, d, partial);
    return launch_status("wn_scale_batch/final");
}
}  // namespace cti

extern "C" size_t cti_wn_scale_workspace_bytes(int n_mats, int64_t elems) {
    if (n_mats <= 0 || elems <= 0) return 0;
    return sizeof(float) * (size_t)n_mats * (size_t)((elems + WN_CHUNK - 1) / WN_CHUNK);
}

extern "C" int cti_wn_scale(const float* weight_v, const float* weight_g, float* scale, int n_mats, int64_t elems, void* workspace,
                            size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(weight_v); CTI_REQUIRE_PTR(weight_g); CTI_REQUIRE_PTR(scale);
    CTI_REQUIRE(n_mats > 0 && elems > 0, CTI_E_SHAPE, "cti_wn_scale: n_mats=%d elems=%lld", n_mats, (long long)elems);
    if (elems <= WN_CHUNK || workspace == nullptr) {       // small matrices: one workgroup each is the fastest form
        hipLaunchKernelGGL(wn_scale_kernel, dim3(n_mats), dim3(1024), 0, as_stream(stream), weight_v, weight_g, scale, elems);
        return launch_status("cti_wn_scale");
    }
    CTI_REQUIRE(workspace_bytes >= cti_wn_scale_workspace_bytes(n_mats, elems), CTI_E_WORKSPACE, "cti_wn_scale: workspace %zu < %zu",
                workspace_bytes, cti_wn_scale_workspace_bytes(n_mats, elems));
    WnBatch wb{};                                           // large matrices: 4,096-element chunks over many CUs, then a fixed-order sum
    wb.n = 1; wb.wv[0] = weight_v; wb.g[0] = weight_g; wb.scale[0] = scale; wb.n_mats[0] = n_mats; wb.elems[0] = elems;
    wn_batch_finish(wb);
    return wn_scale_batch(wb, static_cast<float*>(workspace), as_stream(stream));
}

// scales of MANY layers of different sizes in two launches per WN_MAX layers (a training step recomputes every layer's scale once: 18 launch
// pairs of ~14 us in the FFOE CTI model, 29 in BAN)
extern "C" size_t cti_wn_scale_many_workspace_bytes(const int64_t* elems, int n) {
    size_t chunks = 0;
    for (int i = 0; i < n; ++i) chunks += elems && elems[i] > 0 ? (size_t)((elems[i] + WN_CHUNK - 1) / WN_CHUNK) : 0;
    return sizeof(float) * chunks;
}

extern "C" int cti_wn_scale_many(const float* const* weight_v, const float* const* weight_g, float* const* scale, const int64_t* elems, int n,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(weight_v); CTI_REQUIRE_PTR(weight_g); CTI_REQUIRE_PTR(scale); CTI_REQUIRE_PTR(elems); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(n > 0, CTI_E_SHAPE, "cti_wn_scale_many: n=%d", n);
    for (int i = 0; i < n; ++i)
        CTI_REQUIRE(weight_v[i] && weight_g[i] && scale[i] && elems[i] > 0, CTI_E_SHAPE, "cti_wn_scale_many: entry %d has a NULL pointer or no elements", i);
    CTI_REQUIRE(workspace_bytes >= cti_wn_scale_many_workspace_bytes(elems, n), CTI_E_WORKSPACE, "cti_wn_scale_many: workspace too small");
    float* part = static_cast<float*>(workspace);
    for (int i0 = 0; i0 < n; i0 += WN_MAX) {
        WnBatch wb{};
        wb.n = n - i0 < WN_MAX ? n - i0 : WN_MAX;
        for (int j = 0; j < wb.n; ++j) {
            wb.wv[j] = weight_v[i0 + j]; wb.g[j] = weight_g[i0 + j]; wb.scale[j] = scale[i0 + j]; wb.n_mats[j] = 1; wb.elems[j] = elems[i0 + j];
        }
        wn_batch_finish(wb);
        int rc = wn_scale_batch(wb, part, as_stream(stream)); if (rc) return rc;
        part += wn_batch_partials(wb);
    }
    return 0;
}

extern "C" int cti_zero_row_mask(const float* v, int64_t ldv, uint8_t* mask, int64_t rows, int dim, void* stream) {
    CTI_REQUIRE_PTR(v); CTI_REQUIRE_PTR(mask);
    CTI_REQUIRE(rows > 0 && dim > 0 && ldv >= dim, CTI_E_SHAPE, "cti_zero_row_mask: rows=%lld dim=%d ldv=%lld",
                (long long)rows, dim, (long long)ldv);
    hipLaunchKernelGGL(zero_row_mask_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), reinterpret_cast<const unsigned*>(v), ldv,
                       mask, rows, dim, 0x7fffffffu);
    return launch_status("cti_zero_row_mask");
}

// The same mask of a bf16 matrix (round 5: BASELINE configs[2] / [3] name bf16 tensors): row r is masked iff every element is +-0.  dim and ldv even.
extern "C" int cti_zero_row_mask_bf16(const void* v, int64_t ldv, uint8_t* mask, int64_t rows, int dim, void* stream) {
    CTI_REQUIRE_PTR(v); CTI_REQUIRE_PTR(mask);
    CTI_REQUIRE(rows > 0 && dim > 0 && ldv >= dim, CTI_E_SHAPE, "cti_zero_row_mask_bf16: rows=%lld dim=%d ldv=%lld", (long long)rows, dim, (long long)ldv);
    CTI_REQUIRE((dim & 1) == 0 && (ldv & 1) == 0 && (reinterpret_cast<uintptr_t>(v) & 3) == 0, CTI_E_ALIGN, "cti_zero_row_mask_bf16: dim, ldv must be even and v 4-B aligned");
    hipLaunchKernelGGL(zero_row_mask_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), static_cast<const unsigned*>(v), ldv / 2,
                       mask, rows, dim / 2, 0x7fff7fffu);
    return launch_status("cti_zero_row_mask_bf16");
}

extern "C" int cti_teff_scramble(const float* src, float* dst, int R, int I, int J, int K, int G, int inverse, void* stream) {
    CTI_REQUIRE_PTR(src); CTI_REQUIRE_PTR(dst);
    CTI_REQUIRE(R > 0 && I > 0 && J > 0 && K > 0 && G > 0, CTI_E_SHAPE, "cti_teff_scramble: R=%d I=%d J=%d K=%d G=%d", R, I, J, K, G);
    CTI_REQUIRE(src != dst, CTI_E_SHAPE, "cti_teff_scramble: in-place scramble is not supported");
    const int64_t total = (int64_t)R * I * J * K * G;
    hipLaunchKernelGGL(teff_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), src, dst, J, K, G,
                       total, inverse);
    return launch_status("cti_teff_scramble");
}

namespace cti {
int mbuild_mfma(const float* Vr, const float* Qr, const float* Tt, unsigned short* Mh, unsigned short* Ml, float* Mf, int B, int V, int Q, int R,
                int hr, int G, int64_t pitchM, hipStream_t st);
}

extern "C" int cti_paralind_mbuild_planes_fwd(const float* Vr, const float* Qr, const float* Teff, const float* Teff_t, unsigned short* Mh,
                                              unsigned short* Ml, int B, int V, int Q, int R, int hr, int G, int64_t rows_alloc, void* stream) {
    CTI_REQUIRE_PTR(Vr); CTI_REQUIRE_PTR(Qr); CTI_REQUIRE_PTR(Teff); CTI_REQUIRE_PTR(Mh); CTI_REQUIRE_PTR(Ml);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && R > 0 && hr > 0 && G > 0 && rows_alloc >= (int64_t)B * V * Q * G, CTI_E_SHAPE,
                "cti_paralind_mbuild_planes_fwd: B=%d V=%d Q=%d R=%d hr=%d G=%d rows_alloc=%lld", B, V, Q, R, hr, G, (long long)rows_alloc);
    int rc = CTI_E_UNSUPPORTED;
    if (Teff_t) rc = mbuild_mfma(Vr, Qr, Teff_t, Mh, Ml, nullptr, B, V, Q, R, hr, G, rows_alloc * 16, as_stream(stream));
    if (rc == CTI_E_UNSUPPORTED) rc = mbuild_fast(Vr, Qr, Teff, nullptr, Mh, Ml, B, V, Q, R, hr, G, rows_alloc * 16, as_stream(stream));
    if (rc == CTI_E_UNSUPPORTED) return fail(CTI_E_UNSUPPORTED, "cti_paralind_mbuild_planes_fwd: h/rank=%d G=%d V=%d Q=%d is outside the plane-writing M-build kernels", hr, G, V, Q);
    return rc;
}

namespace cti {
int mbuild_mfma_f6(const float* Vr, const float* Qr, const float* Tt, const F6Planes& P, int B, int V, int Q, int R, int hr, int G, hipStream_t st, const float* Tj = nullptr);
}

extern "C" int cti_paralind_mbuild_f16f6_fwd(const float* Vr, const float* Qr, const float* Teff_t, void* planes, size_t planes_bytes, int B, int V, int Q,
                                             int R, int hr, int G, void* stream) {
    CTI_REQUIRE_PTR(Vr); CTI_REQUIRE_PTR(Qr); CTI_REQUIRE_PTR(Teff_t); CTI_REQUIRE_PTR(planes);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && R > 0 && hr > 0 && G > 0, CTI_E_SHAPE, "cti_paralind_mbuild_f16f6_fwd: B=%d V=%d Q=%d R=%d hr=%d G=%d", B, V, Q, R, hr, G);
    const int64_t rows = (int64_t)B * V * Q * G, rpb = (int64_t)V * Q * G;
    CTI_REQUIRE((R * hr) % 32 == 0, CTI_E_UNSUPPORTED, "cti_paralind_mbuild_f16f6_fwd: R*hr = %d is not a multiple of 32", R * hr);
    CTI_REQUIRE(planes_bytes >= f6_planes_bytes(rows, R * hr, rpb), CTI_E_WORKSPACE, "cti_paralind_mbuild_f16f6_fwd: block %zu < %zu", planes_bytes, f6_planes_bytes(rows, R * hr, rpb));
    CTI_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 255) == 0, CTI_E_ALIGN, "cti_paralind_mbuild_f16f6_fwd: the plane block must be 256-B aligned");
    const int rc = mbuild_mfma_f6(Vr, Qr, Teff_t, f6_carve(planes, rows, R * hr, rpb), B, V, Q, R, hr, G, as_stream(stream));
    if (rc == CTI_E_UNSUPPORTED) return fail(CTI_E_UNSUPPORTED, "cti_paralind_mbuild_f16f6_fwd: h/rank=%d G=%d V=%d Q=%d R=%d is outside the direct-encoding M build (use cti_paralind_mbuild_fwd + cti_quantize_f16f6)", hr, G, V, Q, R);
    return rc;
}

extern "C" int cti_paralind_mbuild_fwd(const float* Vr, const float* Qr, const float* Teff, float* M, int B, int V,
                                       int Q, int R, int I, int J, int K, int G, void* stream) {
    CTI_REQUIRE_PTR(Vr); CTI_REQUIRE_PTR(Qr); CTI_REQUIRE_PTR(Teff); CTI_REQUIRE_PTR(M);
    CTI_REQUIRE(B > 0 && V > 0 && Q > 0 && R > 0 && I > 0 && J > 0 && K > 0 && G > 0, CTI_E_SHAPE,
                "cti_paralind_mbuild_fwd: B=%d V=%d Q=%d R=%d I=%d J=%d K=%d G=%d", B, V, Q, R, I, J, K, G);
    if (I == J && J == K) {                                 // cubic cores (every TCNet): the fast kernel of cti_mbuild.hip, fp32 output
        const int rc = mbuild_fast(Vr, Qr, Teff, M, nullptr, nullptr, B, V, Q, R, I, G, (int64_t)R * K, as_stream(stream));
        if (rc != CTI_E_UNSUPPORTED) return rc;
    }
    const size_t lds = sizeof(float) * ((size_t)J * K * G + I + (size_t)Q * J);
    CTI_REQUIRE(lds <= 64 * 1024, CTI_E_SHAPE, "cti_paralind_mbuild_fwd: J*K*G + I + Q*J = %zu floats exceed 64 KiB of LDS",
                lds / 4);
    hipLaunchKernelGGL(mbuild_kernel, dim3((unsigned)(B * V)), dim3(256), lds, as_stream(stream), Vr, Qr, Teff, M, V, Q, R,
                       I, J, K, G);
    return launch_status("cti_paralind_mbuild_fwd");
}
