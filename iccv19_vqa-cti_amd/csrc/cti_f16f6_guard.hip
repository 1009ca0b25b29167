// cti_f16f6_guard.hip -- range guard of the f16f6 forward (reference src/Tensor.py:12,18: the reference multiplies in full-range fp32).
//
// The f16f6 operand format (cti_f16f6.h) is exact-to-11-bits in its f16 hi part only for 6.1e-5 <= |x| <= 65504: larger values saturate, a
// tensor whose magnitudes sit below the f16 subnormal knee loses its hi part, and the register encoders map NaN to a finite value.  None of
// that may reach a caller silently.  Every encoder already writes, per (row, 32-wide K block), the E8M0 byte of the block's hi maximum into the
// S plane (byte 0 = eh + 127; saturated and non-finite blocks come out as 141 = the byte of 65504), so the guard is ONE small kernel over the
// S planes of everything the f16f6 kernels read -- 2 B per 32 elements, ~75 MB at BASELINE configs[1] -- plus a non-finite scan of the fp32
// rows that feed the direct-encoding M build:
//
//   guard_reset  (first kernel of the call)        zeroes the guard block at the head of the workspace
//   guard_scan   (x2: behind the M build on the auxiliary stream for what chain B and the encoder of `a` produced; before the mode-3 product
//                 for a~ and A^)                   per tensor: max eh byte over its REAL rows (padding rows between batches hold stale bytes);
//                                                   the last workgroup of the FINAL scan evaluates the status word
//   guard_poison (after the mode-3 product)        status != 0: the output is overwritten with NaN -- never a finite, plausible, wrong number
//
// status bits: 1 = a block at or beyond the f16 range (|x| > 61440, incl. inf / NaN in an encoded tensor), 2 = a tensor that is not all zero
// but whose largest block lies below 2^-12 (hi parts subnormal: absolute error 2^-29 stops being small against the tensor), 4 = non-finite
// values in V^ / Q^ / T_eff (the M build's encoder would swallow them), 8 / 16 = the mode-3 product cancels so heavily that the f16f6 (8) or even the
// bf16x3 (16) rounding may exceed 1e-4 of the largest output (guard_cancel's sampled estimate).  The host reads the word behind the verdict's event (cti_guard_read) and
// re-runs the call in the bf16x3 mode; under hipGraph capture, where the host cannot look, the NaN fill is the signal.
#include "cti_common.h"
#include "cti_f16f6.h"

namespace cti {

namespace {

constexpr int GUARD_SAT_BYTE = 141;            // f6_scale_byte(m) for 61440 < m <= 65504 (and what the encoders write for non-finite maxima)
constexpr int GUARD_KNEE_BYTE = 113;           // f6_scale_byte(2^-12)

__global__ __launch_bounds__(64) void guard_reset_kernel(unsigned* words) { words[threadIdx.x] = 0u; }

// What a tensor's scale bytes say, as bits (a slot of the guard block is the OR over the tensor): 1 = a block that is not all zero, 2 = a
// block at or above the 2^-12 knee, 4 = a block at or beyond the f16 range.  fp32 segments: 1 = a non-finite value.
constexpr unsigned GB_NONZERO = 1u, GB_HEALTHY = 2u, GB_SAT = 4u;
__device__ __forceinline__ unsigned guard_bits(unsigned byte) {
    return (byte > 1u ? GB_NONZERO : 0u) | (byte >= (unsigned)GUARD_KNEE_BYTE ? GB_HEALTHY : 0u) | (byte >= (unsigned)GUARD_SAT_BYTE ? GB_SAT : 0u);
}

__global__ __launch_bounds__(256) void guard_scan_kernel(GuardArgs g) {
    __shared__ unsigned wg_bits[GUARD_MAX_SEG];
    if (threadIdx.x < GUARD_MAX_SEG) wg_bits[threadIdx.x] = 0u;
    __syncthreads();
    for (int s = 0; s < g.nseg; ++s) {
        const GuardSeg sg = g.seg[s];
        unsigned bits = 0u;
        if (sg.kind == 0) {
            // S plane: [Kb][rows_allocS][2 B]; real rows = nb batches of rdiv rows (the last one may be shorter: rows_total) starting at multiples
            // of rstride (a multiple of 8).  A workgroup takes whole (K block, batch) pairs -- one scalar division per pair -- and its threads
            // sweep the pair's rows eight at a time with aligned 16-B loads; rows beyond the batch are skipped.
            const unsigned pairs = (unsigned)sg.Kb * (unsigned)sg.nb;
            for (unsigned pr = blockIdx.x; pr < pairs; pr += gridDim.x) {
                const unsigned kb = pr / (unsigned)sg.nb, b = pr - kb * (unsigned)sg.nb;
                const int64_t left_total = sg.rows_total - (int64_t)b * sg.rdiv;
                const int rows = (int)(left_total < sg.rdiv ? left_total : sg.rdiv);
                const uint8_t* base = static_cast<const uint8_t*>(sg.p) + ((int64_t)kb * sg.rows_allocS + (int64_t)b * sg.rstride) * 2;
                for (int r8 = threadIdx.x; r8 * 8 < rows; r8 += 256) {
                    const uint4 w = *reinterpret_cast<const uint4*>(base + (int64_t)r8 * 16);
                    const int left = rows - r8 * 8;                                            // real rows in this item (>= 1)
                    const unsigned d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (left > 2 * u) bits |= guard_bits(d[u] & 0xffu);
                        if (left > 2 * u + 1) bits |= guard_bits((d[u] >> 16) & 0xffu);
                    }
                }
            }
        } else {
            // fp32 rows: any inf / NaN (exponent field all ones)
            const unsigned tid = blockIdx.x * 256u + threadIdx.x, nthr = gridDim.x * 256u;
            const float* x = static_cast<const float*>(sg.p);
            const int64_t n4 = sg.n >> 2;
            for (int64_t i = tid; i < n4; i += nthr) {
                const uint4 w = reinterpret_cast<const uint4*>(x)[i];
                const unsigned e = 0x7f800000u;
                if ((w.x & e) == e || (w.y & e) == e || (w.z & e) == e || (w.w & e) == e) bits = 1u;
            }
            for (int64_t i = (n4 << 2) + tid; i < sg.n; i += nthr)
                if ((__builtin_bit_cast(unsigned, x[i]) & 0x7f800000u) == 0x7f800000u) bits = 1u;
        }
        if (bits) atomicOr(&wg_bits[sg.slot], bits);           // LDS: cheap
    }
    __syncthreads();
    // One global atomic per workgroup and slot AT MOST, and only for bits the word does not show yet: thousands of atomics on one address
    // serialise at the memory side (the first version's per-wave atomicMax cost 0.1 ms).  A stale read only costs a redundant atomicOr.
    if (threadIdx.x < GUARD_MAX_SEG) {
        const unsigned mine = wg_bits[threadIdx.x];
        if (mine) {
            const unsigned seen = __atomic_load_n(&g.words[GUARD_W_SEG + threadIdx.x], __ATOMIC_RELAXED);
            if (mine & ~seen) atomicOr(&g.words[GUARD_W_SEG + threadIdx.x], mine & ~seen);
        }
    }
}

// The verdict, as its OWN one-thread launch behind the final scan (stream order makes every workgroup's bits visible).  Round 4: as a last-arriver pass inside the
// scan it cost one atomic increment of ONE word per workgroup -- 1 024 of them serialise at the memory side: 72 us for a scan of 8 MB, in front of the mode-3 product
// (profiles/r04_step_timeline.txt); the extra launch costs ~5.
__global__ void guard_verdict_kernel(GuardArgs g) {
    if (threadIdx.x != 0) return;
    unsigned status = 0u;
    for (int k = 0; k < g.n_slots; ++k) {
        const unsigned m = atomicOr(&g.words[GUARD_W_SEG + k], 0u);
        if ((g.f32_slots >> k) & 1u) { if (m) status |= CTI_GUARD_NONFINITE; }
        else {
            if (m & GB_SAT) status |= CTI_GUARD_SATURATED;
            if ((m & GB_NONZERO) && !(m & GB_HEALTHY)) status |= CTI_GUARD_UNDERFLOW;
        }
    }
    // cancellation estimate of the mode-3 product (guard_cancel, stream-ordered in front of this scan): the estimated normalised error is
    // ~2^-17 x ratio for the f16f6 product, ~2^-19 x ratio for bf16x3 -- beyond 1e-4 the call belongs in the next mode up
    const float amax = __builtin_bit_cast(float, atomicOr(&g.words[GUARD_W_ABSMAX], 0u)), dmax = __builtin_bit_cast(float, atomicOr(&g.words[GUARD_W_DOTMAX], 0u));
    const float ratio = amax > 0.f ? fminf(amax / fmaxf(dmax, 1e-30f), 3.0e38f) : 0.f;
    atomicExch(&g.words[GUARD_W_RATIO], __builtin_bit_cast(unsigned, ratio));
    if (ratio > g.rho_bf16x3) status |= CTI_GUARD_CANCEL;
    if (ratio > g.rho_fp32) status |= CTI_GUARD_CANCEL_HEAVY;
    atomicExch(&g.words[GUARD_W_STATUS], status);
}

// ---- cancellation estimate (see cti_f16f6.h).  One workgroup per batch: 32 rows of M and 32 rows of A^ are copied from the f16 hi planes into LDS
// (row pitch K * 2 + 16 B: the eight threads that share an M row read eight different A^ rows without a bank conflict), thread t multiplies the
// pairs (t >> 3, (t & 7) + 8 u), u < 4, with v_dot2_f32_f16 -- once as they stand, once with the sign bits cleared: 1 024 pairs per batch.
// WHICH rows (round 5; VERDICT r4: cancellation confined to a few answer tokens was invisible to 32 evenly spaced rows of 3 129): 16 evenly spaced rows
// (offset by the batch index) + the LARGEST row of each of 16 contiguous strata of the operand's rows, "largest" by the sum over the K blocks of
// 2^eh from the scale bytes every encoder has written already (2 B per row and block: 100 KB per sample for A^ at BASELINE configs[1]) -- the pair with
// the largest sum_k |m_k a_k| is a pair of large rows, and a handful of outsized answer tokens are each the maximum of their stratum (ties: lowest row;
// 64-bit LDS atomicMax on (proxy bits, ~row): deterministic).
typedef _Float16 gc_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned gc_u32x4 __attribute__((ext_vector_type(4)));
constexpr int GC_ROWS = 32, GC_EVEN = 16, GC_STRATA = GC_ROWS - GC_EVEN;
__device__ __forceinline__ void gc_strata_max(const char* S, int64_t ras, int64_t row0, int rows, int Kb, int kb0, int kstep, unsigned long long* best, int t) {
    // thread t owns the 8-row groups t, t + 256, ...: one aligned 16-B load per (group, K block); deep contractions look at every kstep-th block
    // (an outsized row is outsized in all of its blocks; the kernel sits in front of the mode-3 product: 25 KB instead of 100 KB per sample)
    for (int g8 = t; g8 * 8 < rows; g8 += 256) {
        float px[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int kb = kb0; kb < Kb; kb += kstep) {
            const gc_u32x4 w = *reinterpret_cast<const gc_u32x4*>(S + ((int64_t)kb * ras + row0 + (int64_t)g8 * 8) * 2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                px[2 * u] += __builtin_bit_cast(float, (w[u] & 0xffu) << 23);               // 2^(eh): byte 0 of the row's (hi, lo) pair
                px[2 * u + 1] += __builtin_bit_cast(float, ((w[u] >> 16) & 0xffu) << 23);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = g8 * 8 + u;
            if (row >= rows) break;
            const float pv = px[u] < 3.0e38f ? px[u] : 3.0e38f;
            const unsigned long long key = ((unsigned long long)__builtin_bit_cast(unsigned, pv) << 32) | (unsigned long long)(0xffffffffu - (unsigned)row);
            atomicMax(&best[(int)((int64_t)row * GC_STRATA / rows)], key);
        }
    }
}
__global__ __launch_bounds__(256) void guard_cancel_kernel(const char* MH, const char* MS, int64_t m_ra, int64_t m_ras, int64_t m_rstride, int mrows, const char* AH,
                                                            const char* AS, int64_t a_ra, int64_t a_ras, int64_t a_rstride, int arows, int Kb, unsigned* words, int strata) {
    extern __shared__ __attribute__((aligned(16))) char gc_lds[];
    __shared__ float red[2][4];
    __shared__ unsigned long long best[2][GC_STRATA];
    __shared__ int sel[2][GC_ROWS];
    const int b = blockIdx.x, t = threadIdx.x;
    const int pitch = Kb * 64 + 16;
    if (t < 2 * GC_STRATA) best[t / GC_STRATA][t % GC_STRATA] = 0ull;
    __syncthreads();
    const int kstep = Kb >= 8 ? 4 : 1, kb0 = b % kstep;
    if (strata) {
        gc_strata_max(MS, m_ras, (int64_t)b * m_rstride, mrows, Kb, kb0, kstep, best[0], t);
        gc_strata_max(AS, a_ras, (int64_t)b * a_rstride, arows, Kb, kb0, kstep, best[1], t);
    }
    __syncthreads();
    if (t < 2 * GC_ROWS) {
        const int op = t / GC_ROWS, i = t % GC_ROWS;
        const int rows = op ? arows : mrows;
        int r = (int)(((int64_t)(i % GC_EVEN) * rows / GC_EVEN + (op ? 11 : 5) * (int64_t)b) % rows);       // evenly spaced, offset by the batch index
        if (!strata) r = (int)(((int64_t)i * rows / GC_ROWS + (op ? 11 : 5) * (int64_t)b) % rows);           // (test knob: round 4's 32 evenly spaced rows)
        else if (i >= GC_EVEN) {
            const unsigned long long k = best[op][i - GC_EVEN];
            if (k != 0ull) r = (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));      // (an empty stratum -- fewer rows than strata -- keeps the even row)
        }
        sel[op][i] = r;
    }
    __syncthreads();
    // copy: 64 rows x Kb blocks of 64 B; thread -> (row, block, 16-B piece)
    for (int it = t; it < 2 * GC_ROWS * Kb * 4; it += 256) {
        const int piece = it & 3, kb = (it >> 2) % Kb, row = (it >> 2) / Kb;
        const bool isA = row >= GC_ROWS;
        const int64_t r = sel[isA ? 1 : 0][row & (GC_ROWS - 1)];
        const char* src = isA ? AH + (((int64_t)kb * a_ra + (int64_t)b * a_rstride + r) * 64 + piece * 16)
                              : MH + (((int64_t)kb * m_ra + (int64_t)b * m_rstride + r) * 64 + piece * 16);
        *reinterpret_cast<gc_u32x4*>(gc_lds + row * pitch + kb * 64 + piece * 16) = *reinterpret_cast<const gc_u32x4*>(src);
    }
    __syncthreads();
    const char* mr = gc_lds + (t >> 3) * pitch;
    const char* ar = gc_lds + (GC_ROWS + (t & 7)) * pitch;
    float dot[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Kb * 4; ++c) {
        const gc_u32x4 m = *reinterpret_cast<const gc_u32x4*>(mr + c * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const gc_u32x4 a = *reinterpret_cast<const gc_u32x4*>(ar + u * 8 * pitch + c * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dot[u] = __builtin_amdgcn_fdot2(__builtin_bit_cast(gc_f16x2, m[e]), __builtin_bit_cast(gc_f16x2, a[e]), dot[u], false);
                ab[u] = __builtin_amdgcn_fdot2(__builtin_bit_cast(gc_f16x2, m[e] & 0x7fff7fffu), __builtin_bit_cast(gc_f16x2, a[e] & 0x7fff7fffu), ab[u], false);
            }
        }
    }
    float md = wave_max(fmaxf(fmaxf(fabsf(dot[0]), fabsf(dot[1])), fmaxf(fabsf(dot[2]), fabsf(dot[3]))));
    float ma = wave_max(fmaxf(fmaxf(ab[0], ab[1]), fmaxf(ab[2], ab[3])));
    if ((t & 63) == 0) { red[0][t >> 6] = md; red[1][t >> 6] = ma; }
    __syncthreads();
    if (t == 0) {
        md = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        ma = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        // non-finite sums are the range bits' business; an all-zero batch (ma == 0) has nothing to cancel.  Both maxima are over the whole call (positive
        // floats order like their bit patterns); one atomic per batch and word at most, and only when it would raise the word
        if (ma > 0.f && ma < 3.0e38f) {
            const unsigned ba = __builtin_bit_cast(unsigned, ma), bd = __builtin_bit_cast(unsigned, fminf(md, 3.0e38f));
            if (ba > __atomic_load_n(&words[GUARD_W_ABSMAX], __ATOMIC_RELAXED)) atomicMax(&words[GUARD_W_ABSMAX], ba);
            if (bd > __atomic_load_n(&words[GUARD_W_DOTMAX], __ATOMIC_RELAXED)) atomicMax(&words[GUARD_W_DOTMAX], bd);
        }
    }
}

__global__ __launch_bounds__(256) void guard_poison_kernel(const unsigned* words, float* out, int64_t n, unsigned bits) {
    if ((words[GUARD_W_STATUS] & bits) == 0u) return;
    const float nan = __builtin_nanf("");
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = nan;
}

}  // namespace

int guard_reset(unsigned* words, hipStream_t st) {
    hipLaunchKernelGGL(guard_reset_kernel, dim3(1), dim3(64), 0, st, words);
    return launch_status("guard_reset");
}

int guard_scan(const GuardArgs& g, hipStream_t st) {
    if (g.nseg <= 0 || g.nseg > GUARD_MAX_SEG) return fail(CTI_E_SHAPE, "guard_scan: %d segments", g.nseg);
    for (int s = 0; s < g.nseg; ++s) {
        const GuardSeg& sg = g.seg[s];
        if (sg.slot < 0 || sg.slot >= GUARD_MAX_SEG) return fail(CTI_E_SHAPE, "guard_scan: segment %d has slot %d", s, sg.slot);
        if (sg.kind == 0 && ((sg.rstride & 7) || (sg.rows_allocS & 7) || sg.rdiv <= 0 || sg.nb <= 0 || sg.rdiv >= (1ll << 31) ||
                             (int64_t)sg.Kb * sg.nb >= (1ll << 32) || (reinterpret_cast<uintptr_t>(sg.p) & 15)))
            return fail(CTI_E_SHAPE, "guard_scan: segment %d (Kb=%d nb=%lld rdiv=%lld rstride=%lld)", s, sg.Kb, (long long)sg.nb, (long long)sg.rdiv, (long long)sg.rstride);
        if (sg.kind == 1 && (reinterpret_cast<uintptr_t>(sg.p) & 15)) return fail(CTI_E_ALIGN, "guard_scan: fp32 segment %d is not 16-B aligned", s);
    }
    hipLaunchKernelGGL(guard_scan_kernel, dim3(1024), dim3(256), 0, st, g);
    if (g.final) {
        GuardArgs gv = g;
        gv.rho_bf16x3 = tuning_guard_rho(0); gv.rho_fp32 = tuning_guard_rho(1);
        hipLaunchKernelGGL(guard_verdict_kernel, dim3(1), dim3(64), 0, st, gv);
    }
    return launch_status("guard_scan");
}

int guard_cancel(const F6Planes& M, int64_t mrows, const F6Planes& A, int64_t arows, int nb, unsigned* words, hipStream_t st) {
    if (M.Kb != A.Kb || M.Kb <= 0 || M.Kb > 32 || mrows <= 0 || arows <= 0 || nb <= 0) return CTI_OK;       // (deep contractions: no estimate, the range bits still hold)
    const int lds = 2 * GC_ROWS * (M.Kb * 64 + 16);
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(guard_cancel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GC_ROWS * (32 * 64 + 16));
        if (e != hipSuccess) return fail((int)e, "guard_cancel: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    hipLaunchKernelGGL(guard_cancel_kernel, dim3((unsigned)nb), dim3(256), lds, st, reinterpret_cast<const char*>(M.H), reinterpret_cast<const char*>(M.S), M.rows_alloc,
                       M.rows_allocS, nb > 1 ? M.rstride : 0, (int)mrows, reinterpret_cast<const char*>(A.H), reinterpret_cast<const char*>(A.S), A.rows_alloc, A.rows_allocS,
                       nb > 1 ? A.rstride : 0, (int)arows, M.Kb, words, tuning_guard_strata());
    return launch_status("guard_cancel");
}

int guard_poison(const unsigned* words, float* out, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(guard_poison_kernel, dim3(2048), dim3(256), 0, st, words, out, n, tuning_guard_poison_bits());
    return launch_status("guard_poison");
}

}  // namespace cti

using namespace cti;

extern "C" int cti_guard_read_ratio(const void* workspace, void* stream, float* ratio_host) {
    CTI_REQUIRE_PTR(workspace); CTI_REQUIRE_PTR(ratio_host);
    hipError_t e = hipMemcpyAsync(ratio_host, static_cast<const unsigned*>(workspace) + GUARD_W_RATIO, sizeof(float), hipMemcpyDeviceToHost, as_stream(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    return e == hipSuccess ? CTI_OK : fail((int)e, "cti_guard_read_ratio: %s", hipGetErrorString(e));
}

extern "C" int cti_guard_read(const void* workspace, void* ev_core_begin, void* stream, uint32_t* status_host) {
    CTI_REQUIRE_PTR(workspace); CTI_REQUIRE_PTR(ev_core_begin); CTI_REQUIRE_PTR(status_host);
    hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(ev_core_begin));       // the scan precedes the event on the launch stream
    if (e == hipSuccess) e = hipMemcpyAsync(status_host, static_cast<const unsigned*>(workspace) + GUARD_W_STATUS, sizeof(uint32_t), hipMemcpyDeviceToHost, as_stream(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
    return e == hipSuccess ? CTI_OK : fail((int)e, "cti_guard_read: %s", hipGetErrorString(e));
}
