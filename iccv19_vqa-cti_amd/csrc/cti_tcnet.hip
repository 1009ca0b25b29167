// cti_tcnet.hip -- TCNet.forward (reference src/tc.py:41-52) as ONE C-ABI call: the whole launch sequence with the
// intermediates chained in the layout the next kernel consumes (bf16 hi/lo planes between MFMA GEMMs, fp32 where a
// VALU kernel reads them), so that no intermediate makes an extra fp32 round trip through HBM.
//
//   scales (weight-norm) -> [split inputs + weights] -> 3 Tucker GEMMs -> 3 packed rank GEMMs -> T_eff scramble
//   -> M build (modes 1+2) -> mode-3 GEMM + rank sum, written as out[b,v,q,a,g]
#include "cti_common.h"
#include "cti_f16f6.h"

namespace cti {
int mbuild_mfma(const float* Vr, const float* Qr, const float* Tt, unsigned short* Mh, unsigned short* Ml, float* Mf, int B, int V, int Q, int R,
                int hr, int G, int64_t pitchM, hipStream_t st);
int mbuild_mfma_f6(const float* Vr, const float* Qr, const float* Tt, const F6Planes& P, int B, int V, int Q, int R, int hr, int G, hipStream_t st, const float* Tj = nullptr);
int mbuild_f6_tj_layout(const float* Teff, float* Tj, int R, int hr, int G, hipStream_t st);
int core_small_tk_layout(const float* Teff, float* Tk, int R, int hr, int G, hipStream_t st);
int mbuild_core_small(const float* Vr, const float* Qr, const float* Tt, const float* Ar, float* out, int B, int V, int Q, int A, int R, int hr, int G,
                      hipStream_t st, const uint8_t* sm_mask = nullptr, float* sm_p = nullptr, int v_rep = 1, int terms = 3, const float* Tk = nullptr);
int mbuild_fast(const float* Vr, const float* Qr, const float* Teff, float* Mf, unsigned short* Mh, unsigned short* Ml, int B,
                int V, int Q, int R, int hr, int G, int64_t ldm_or_pitch, hipStream_t st);
bool mbuild_mfma_f6_fits(int B, int V, int Q, int R, int hr, int G);              // the launchers' own shape tests (cti_mbuild.hip), by sizes only
bool mbuild_mfma_fits(int B, int V, int Q, int R, int hr, int G);
bool mbuild_fast_fits(int B, int V, int Q, int R, int hr, int G);
bool mbuild_core_small_fits(int B, int V, int Q, int A, int R, int hr, int G);
}
using namespace cti;

#ifndef CTI_AF32
#define CTI_AF32 1       // fp32 A operand of the Tucker GEMMs split at fragment-read time instead of a split pass: -2.6 % on the whole step once the
#endif                   // LDS reads of that path stopped draining vmcnt(0) (plain vector type instead of HIP's float4 struct); 0 = the split pass

namespace {

// the Tucker GEMM of side s reads its fp32 input directly (no input planes in the workspace): decided by sizes only, so that
// cti_tcnet_forward_workspace_bytes and the call agree; the call then REQUIRES a 16-B aligned input pointer
inline bool af32_side(int prec, int in_dim) { return CTI_AF32 && (prec == CTI_PREC_BF16X3 || prec == CTI_PREC_F16F6) && in_dim % 4 == 0; }

struct Bump {
    char* base; size_t off, cap;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

struct Planes { unsigned short* hi; unsigned short* lo; int Kp; int64_t rows_alloc; };
Planes take_planes(Bump& w, int64_t rows, int K) {
    const int Kp = planes_kp(K);
    Planes p;
    p.rows_alloc = rows + PLANE_SLACK_ROWS;
    const size_t n = (size_t)p.rows_alloc * Kp;
    p.hi = static_cast<unsigned short*>(w.take(2 * sizeof(unsigned short) * n));
    p.lo = p.hi ? p.hi + n : nullptr;
    p.Kp = Kp;
    return p;
}

struct Dims { int B, V, Q, A, vd, qd, ad, h, R, G; };

// Few answer tokens (the FFOE / MC models: A = 3 / 6): modes 1 + 2 + 3 run in ONE kernel (mbuild_core_small) and M is never written.
// The kernel's own shape test (hr = 16, glimpse 2, V <= 64, Q <= 16, A <= 6, even rank count AND its LDS budget: X + the sample's A^ block --
// with R = 32 that refuses V >= 62 at A = 3 and V >= 60 at A = 6, which then take the M build + planes GEMM).
static bool small_a(const Dims& d) { return mbuild_core_small_fits(d.B, d.V, d.Q, d.A, d.R, d.h / d.R, d.G); }

// f16f6 mode: the M build encodes the mode-3 product's planes itself (mbuild_mfma_f6's own shape test, by sizes only -- the workspace then
// holds no fp32 M); other shapes build fp32 rows and run the encoding pass.
static bool direct_m(const Dims& d) { return mbuild_mfma_f6_fits(d.B, d.V, d.Q, d.R, d.h / d.R, d.G); }
// bf16 planes modes: neither plane-writing M build takes the shape (their LDS budgets end at ~56 objects for hr = 16; other hr / odd rank counts
// never fit): the generic kernel builds fp32 rows that a split pass turns into planes -- in `out` when it is large enough (A >= h), else in a scratch block
static bool m_needs_scratch(const Dims& d) {
    const int hr = d.h / d.R;
    return !mbuild_mfma_fits(d.B, d.V, d.Q, d.R, hr, d.G) && !mbuild_fast_fits(d.B, d.V, d.Q, d.R, hr, d.G) && d.A < d.h;
}

// One pass over the carve plan: with base == nullptr it only measures.
struct Plan {
    float* scale_t[3]; float* scale_r[3]; float* Teff; float* Tt; float* Tk; float* Tj; float* wn_partial;
    // fp32 mode
    float* t32[3]; float* r32[3]; float* M32;
    // planes mode
    Planes xin[3], wt[3], wr[3], tp[3], Arp, Mp;
    float* Vr; float* Qr;
    float* Ar32;                               // few answer tokens (small_a): A^ as fp32 rows for the fused modes-1+2+3 kernel; no M, no A^ planes
    // f16f6 mode: the mode-3 product runs on f16 + fp6 planes (cti_f16f6.h) written by the rank GEMM's epilogue (A^) and an encoding pass (M);
    // every other GEMM stays on bf16x3
    F6Planes f_Arp, f_Mp; float* Mf32;
    float* Mscr;                               // bf16 planes modes, m_needs_scratch(): fp32 M rows of the generic M build
    // ... and so do the a-side rank nets (A >= 7): the Tucker GEMM's epilogue encodes a~ (f_At), the rank nets run as one transposed f16f6 product
    // whose register epilogue encodes A^ (gemm_nt_f16f6 epi 6) against the rank weights' block f_wra (weight-norm scale folded in; prepared)
    F6Planes f_At, f_wra;
    // ... and the a-side Tucker projection: `a` is encoded once (f_Ain) and multiplied with the Tucker weight's block f_wta (scale folded in; prepared)
    // in the same transposed form
    F6Planes f_Ain, f_wta;
    unsigned* guard;                           // f16f6 kernels in use: the range-guard block (GUARD_WORDS uint32) at the HEAD of the per-call workspace
    size_t bytes;
};

// The batch-independent part of the plan -- weight-norm scales, T_eff (+ its transposed copy), the weights' operand planes: carved
// either at the head of the per-call workspace (and recomputed every call) or in a caller-kept "prepared" block (cti_tcnet_prepare:
// computed once per parameter update, the way inference holds its weights).
void carve_prep(const Dims& d, int prec, Bump& w, Plan& p) {
    const int hr = d.h / d.R;
    const int in[3] = {d.vd, d.qd, d.ad};
    for (int s = 0; s < 3; ++s) { p.scale_t[s] = static_cast<float*>(w.take(sizeof(float))); p.scale_r[s] = static_cast<float*>(w.take(sizeof(float) * d.R)); }
    p.Teff = static_cast<float*>(w.take(sizeof(float) * (size_t)d.R * hr * hr * hr * d.G));
    p.Tt = static_cast<float*>(w.take(sizeof(float) * (size_t)d.R * hr * hr * hr * d.G));      // [r][c][i]: the MFMA M build's B operand
    p.Tk = static_cast<float*>(w.take(sizeof(float) * (size_t)d.R * hr * hr * hr * d.G));      // [(r, i)][g][j][k]: the round-6 few-answer core's A operand (k contiguous)
    p.Tj = static_cast<float*>(w.take(sizeof(float) * (size_t)d.R * hr * hr * hr * d.G));      // [(r, k)][g][j][i]: the round-6 direct-encoding M build's A operand (i contiguous)
    {
        size_t chunks = 0;
        for (int s = 0; s < 3; ++s) chunks += (size_t)(((int64_t)d.h * in[s] + WN_CHUNK - 1) / WN_CHUNK) + (size_t)d.R * (((int64_t)hr * d.h + WN_CHUNK - 1) / WN_CHUNK);
        p.wn_partial = static_cast<float*>(w.take(sizeof(float) * chunks));
    }
    if (prec != CTI_PREC_F32)
        for (int s = 0; s < 3; ++s) { p.wt[s] = take_planes(w, d.h, in[s]); p.wr[s] = take_planes(w, d.h, d.h); }
    if (prec == CTI_PREC_F16F6) {
        p.f_wra = f6_carve(w.take(f6_planes_bytes(d.h, d.h, 0)), d.h, d.h, 0);
        p.f_wta = f6_carve(w.take(f6_planes_bytes(d.h, d.ad, 0)), d.h, d.ad, 0);
    }
}

Plan carve(const Dims& d, int prec, void* ws) {
    Plan p{};
    Bump w{static_cast<char*>(ws), 0, 0};
    const int64_t rows[3] = {(int64_t)d.B * d.V, (int64_t)d.B * d.Q, (int64_t)d.B * d.A};
    const int in[3] = {d.vd, d.qd, d.ad};
    if (prec == CTI_PREC_F16F6 && !small_a(d)) p.guard = static_cast<unsigned*>(w.take(sizeof(unsigned) * GUARD_WORDS));     // offset 0: include/cti_hip.h
    carve_prep(d, prec, w, p);
    const int64_t mrows = (int64_t)d.B * d.V * d.Q * d.G;
    if (prec == CTI_PREC_F32) {
        for (int s = 0; s < 3; ++s) {
            p.t32[s] = static_cast<float*>(w.take(sizeof(float) * rows[s] * d.h));
            p.r32[s] = static_cast<float*>(w.take(sizeof(float) * rows[s] * d.h));
        }
        p.M32 = static_cast<float*>(w.take(sizeof(float) * mrows * d.h));
    } else {
        const bool f6 = prec == CTI_PREC_F16F6;
        for (int s = 0; s < 3; ++s) {
            if (s == 2 && f6 && !small_a(d)) p.xin[s] = Planes{};
            else if (!af32_side(prec, in[s])) p.xin[s] = take_planes(w, rows[s], in[s]);
            else { p.xin[s] = Planes{}; p.xin[s].Kp = planes_kp(in[s]); p.xin[s].rows_alloc = rows[s] + PLANE_SLACK_ROWS; }
            if (s == 2 && f6 && !small_a(d)) {
                p.f_Ain = f6_carve(w.take(f6_planes_bytes(rows[2], d.ad, 0)), rows[2], d.ad, 0);
                p.f_At = f6_carve(w.take(f6_planes_bytes(rows[2], d.h, 0)), rows[2], d.h, 0);
            }
            else p.tp[s] = take_planes(w, rows[s], d.h);
        }
        p.Vr = static_cast<float*>(w.take(sizeof(float) * rows[0] * d.h));
        p.Qr = static_cast<float*>(w.take(sizeof(float) * rows[1] * d.h));
        if (small_a(d)) {
            p.Ar32 = static_cast<float*>(w.take(sizeof(float) * rows[2] * d.h));
        } else if (!f6) {
            p.Arp = take_planes(w, rows[2], d.h);
            p.Mp = take_planes(w, mrows, d.h);
            if (m_needs_scratch(d)) p.Mscr = static_cast<float*>(w.take(sizeof(float) * mrows * d.h));
        } else {
            const int64_t mpb = (int64_t)d.V * d.Q * d.G;
            p.f_Arp = f6_carve(w.take(f6_planes_bytes(rows[2], d.h, d.A)), rows[2], d.h, d.A);
            p.Mf32 = direct_m(d) ? nullptr : static_cast<float*>(w.take(sizeof(float) * mrows * d.h));
            p.f_Mp = f6_carve(w.take(f6_planes_bytes(mrows, d.h, mpb)), mrows, d.h, mpb);
        }
    }
    p.bytes = (w.off + 255) & ~(size_t)255;
    return p;
}

int check_dims(const Dims& d) {
    CTI_REQUIRE(d.B > 0 && d.V > 0 && d.Q > 0 && d.A > 0 && d.vd > 0 && d.qd > 0 && d.ad > 0 && d.h > 0 && d.R > 0 && d.G > 0, CTI_E_SHAPE,
                "cti_tcnet_forward: B=%d V=%d Q=%d A=%d v_dim=%d q_dim=%d a_dim=%d h=%d R=%d G=%d", d.B, d.V, d.Q, d.A, d.vd, d.qd, d.ad, d.h, d.R, d.G);
    CTI_REQUIRE(d.h % d.R == 0, CTI_E_SHAPE, "cti_tcnet_forward: h=%d is not a multiple of rank=%d", d.h, d.R);
    return CTI_OK;
}

// the a-side rank weights (R matrices of hr x h = one h x h matrix) and Tucker weight as f16f6 blocks with their weight-norm scales folded in;
// the blocks' slack rows are zeroed (defined scales, zero codes)
int quantize_rank_a(const Dims& d, const Plan& p, const float* rank_wv_a, const float* tucker_wv_a, hipStream_t st) {
    hipError_t e = hipMemsetAsync(p.f_wra.H, 0, f6_planes_bytes(d.h, d.h, 0), st);      // (H is the block's first plane)
    if (e == hipSuccess) e = hipMemsetAsync(p.f_wta.H, 0, f6_planes_bytes(d.h, d.ad, 0), st);
    if (e != hipSuccess) return fail((int)e, "cti_tcnet: hipMemsetAsync: %s", hipGetErrorString(e));
    int rc = quantize_f16f6(rank_wv_a, d.h, d.h, d.h, p.f_wra, st, p.scale_r[2], d.h / d.R); if (rc) return rc;
    return quantize_f16f6(tucker_wv_a, d.ad, d.h, d.ad, p.f_wta, st, p.scale_t[2], d.h);
}

// scales of the six weight-normalised layers, T_eff (and its transposed copy), and -- planes modes -- the weights' operand planes
int run_prepare(const Dims& d, int prec, const Plan& p, const float* const* tucker_wv, const float* const* tucker_g, const float* const* rank_wv,
                const float* const* rank_g, const float* T_g, bool weight_planes, void* stream) {
    const int hr = d.h / d.R;
    const int in[3] = {d.vd, d.qd, d.ad};
    hipStream_t st = as_stream(stream);
    WnBatch wb{};
    wb.n = 6;
    for (int s = 0; s < 3; ++s) {
        wb.wv[s] = tucker_wv[s]; wb.g[s] = tucker_g[s]; wb.scale[s] = p.scale_t[s]; wb.n_mats[s] = 1; wb.elems[s] = (int64_t)d.h * in[s];
        wb.wv[3 + s] = rank_wv[s]; wb.g[3 + s] = rank_g[s]; wb.scale[3 + s] = p.scale_r[s]; wb.n_mats[3 + s] = d.R; wb.elems[3 + s] = (int64_t)hr * d.h;
    }
    wn_batch_finish(wb);
    int rc = wn_scale_batch(wb, p.wn_partial, st); if (rc) return rc;
    rc = cti_teff_scramble(T_g, p.Teff, d.R, hr, hr, hr, d.G, 0, stream); if (rc) return rc;
    if (prec != CTI_PREC_F32) {                                 // T_eff[r] (i x c) -> Tt[r] (c x i): contraction axis contiguous
        rc = cti_transpose_f32(p.Teff, (int64_t)hr * hr * d.G, (int64_t)hr * hr * hr * d.G, p.Tt, hr, (int64_t)hr * hr * hr * d.G, hr, hr * hr * d.G, d.R, stream);
        if (rc) return rc;
        rc = core_small_tk_layout(p.Teff, p.Tk, d.R, hr, d.G, st); if (rc) return rc;
        rc = mbuild_f6_tj_layout(p.Teff, p.Tj, d.R, hr, d.G, st); if (rc) return rc;
        if (weight_planes) {
            for (int s = 0; s < 3; ++s) {
                rc = split_planes(tucker_wv[s], in[s], d.h, in[s], p.wt[s].hi, p.wt[s].lo, p.wt[s].rows_alloc, st); if (rc) return rc;
                rc = split_planes(rank_wv[s], d.h, d.h, d.h, p.wr[s].hi, p.wr[s].lo, p.wr[s].rows_alloc, st); if (rc) return rc;
            }
            if (prec == CTI_PREC_F16F6) { rc = quantize_rank_a(d, p, rank_wv[2], tucker_wv[2], st); if (rc) return rc; }
        }
    }
    return CTI_OK;
}

}  // namespace

extern "C" size_t cti_tcnet_prepared_bytes(int v_dim, int q_dim, int a_dim, int h, int R, int G, int prec) {
    if (v_dim <= 0 || q_dim <= 0 || a_dim <= 0 || h <= 0 || R <= 0 || G <= 0 || h % R) return 0;
    Dims d{1, 1, 1, 1, v_dim, q_dim, a_dim, h, R, G};
    Plan p{};
    Bump w{nullptr, 0, 0};
    carve_prep(d, prec, w, p);
    return (w.off + 255) & ~(size_t)255;
}

extern "C" int cti_tcnet_prepare(const float* const* tucker_wv, const float* const* tucker_g, const float* const* rank_wv, const float* const* rank_g,
                                 const float* T_g, int v_dim, int q_dim, int a_dim, int h, int R, int G, int prec, void* prepared, size_t prepared_bytes,
                                 void* stream) {
    CTI_REQUIRE_PTR(tucker_wv); CTI_REQUIRE_PTR(tucker_g); CTI_REQUIRE_PTR(rank_wv); CTI_REQUIRE_PTR(rank_g); CTI_REQUIRE_PTR(T_g); CTI_REQUIRE_PTR(prepared);
    Dims d{1, 1, 1, 1, v_dim, q_dim, a_dim, h, R, G};
    int rc = check_dims(d); if (rc) return rc;
    CTI_REQUIRE(prec == CTI_PREC_F32 || prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16 || prec == CTI_PREC_F16F6, CTI_E_UNSUPPORTED, "cti_tcnet_prepare: prec=%d", prec);
    CTI_REQUIRE(prec != CTI_PREC_F16F6 || h % 32 == 0, CTI_E_UNSUPPORTED, "cti_tcnet_prepare: the f16f6 mode needs h %% 32 == 0 (h=%d)", h);
    CTI_REQUIRE(prepared_bytes >= cti_tcnet_prepared_bytes(v_dim, q_dim, a_dim, h, R, G, prec), CTI_E_WORKSPACE, "cti_tcnet_prepare: block too small");
    for (int s = 0; s < 3; ++s)
        CTI_REQUIRE(tucker_wv[s] && tucker_g[s] && rank_wv[s] && rank_g[s], CTI_E_NULL, "cti_tcnet_prepare: weight pointer %d is NULL", s);
    Plan p{};
    Bump w{static_cast<char*>(prepared), 0, 0};
    carve_prep(d, prec, w, p);
    return run_prepare(d, prec, p, tucker_wv, tucker_g, rank_wv, rank_g, T_g, true, stream);
}

extern "C" size_t cti_tcnet_forward_workspace_bytes(int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                                                    int G, int prec) {
    Dims d{B, V, Q, A, v_dim, q_dim, a_dim, h, R, G};
    if (B <= 0 || V <= 0 || Q <= 0 || A <= 0 || v_dim <= 0 || q_dim <= 0 || a_dim <= 0 || h <= 0 || R <= 0 || G <= 0 || h % R) return 0;
    return carve(d, prec, nullptr).bytes;
}

extern "C" size_t cti_tcnet_forward_guard_bytes(int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R, int G, int prec) {
    if (B <= 0 || V <= 0 || Q <= 0 || A <= 0 || v_dim <= 0 || q_dim <= 0 || a_dim <= 0 || h <= 0 || R <= 0 || G <= 0 || h % R || prec != CTI_PREC_F16F6 || h % 32) return 0;
    Dims d{B, V, Q, A, v_dim, q_dim, a_dim, h, R, G};
    return small_a(d) ? 0 : sizeof(unsigned) * GUARD_WORDS;
}

// CTI_F6_GUARD_ABLATE=1 (measurement only): the guard's scan / estimate / poison kernels are not launched (the block is still reset, so the host reads
// status 0): the guard-on vs guard-off A/B of the step, profiles/r04_guard_ab.txt
static bool guard_ablate() { static const bool v = [] { const char* e = getenv("CTI_F6_GUARD_ABLATE"); return e && e[0] == '1'; }(); return v; }

// Guard scheduling.  Round 3 spread the scans so that only M's 8 MB of scale bytes sat in front of the mode-3 product: `a`, a~, the weights and the fp32 sweep
// on the auxiliary stream (behind an event wait for the a side's Tucker product), A^ behind the rank product.  The round-4 A/B (profiles/r04_guard_ab.txt)
// priced the guard at 0.15-0.19 ms per step: 1 024-workgroup scans beside persistent GEMMs that hold every CU wait for the GEMM to end, and chain B -- whose
// M build IS the tail of the critical path -- waited for an event of the main stream.  CTI_F6_GUARD_LATE=1 (experiment): ONE scan of everything (~100 MB at
// configs[1]) behind the join, unstarved, in front of the mode-3 product -- measured SLOWER than the spread placement (4.52-4.54 vs 4.45-4.50 ms per step,
// no guard kernels at all: 4.34-4.47; profiles/r04_guard_ab.txt): the spread scans do hide, what shows of the guard is ~0.1 ms.
static bool guard_late() { static const bool v = [] { const char* e = getenv("CTI_F6_GUARD_LATE"); return e && e[0] == '1'; }(); return v; }

static bool sm_partials_supported(int h, int G, int prec) { return prec == CTI_PREC_F16F6 && G == 2 && h % 32 == 0; }

extern "C" size_t cti_tcnet_softmax_partials_bytes(int B, int V, int Q, int A, int h, int G, int prec) {
    if (B <= 0 || V <= 0 || Q <= 0 || A <= 6 || !sm_partials_supported(h, G, prec)) return 0;      // (A <= 6 may take the fused modes-1+2+3 kernel, which leaves no partials)
    return sizeof(float) * (size_t)B * f6_sm_chunks(V * Q * G, A) * G * 2;
}

static int tcnet_forward_impl(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                              const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                              const float* const* rank_g, const float* const* rank_b, const float* T_g, float* out,
                              uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                              int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                              void* ev_core_end, void* aux_stream, void* stream, float* sm_part, float* p_fused = nullptr,
                              const float* v_tucked = nullptr, int64_t ld_vt = 0, int v_rep = 1, int v16 = 0);

extern "C" int cti_tcnet_forward(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                                 const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                                 const float* const* rank_g, const float* const* rank_b, const float* T_g, float* out,
                                 uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                                 int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                                 void* ev_core_end, void* aux_stream, void* stream) {
    return tcnet_forward_impl(v, q, a, tucker_wv, tucker_g, tucker_b, rank_wv, rank_g, rank_b, T_g, out, zero_mask, B, V, Q, A, v_dim, q_dim, a_dim, h, R,
                              G, act, prec, prepared, workspace, workspace_bytes, ev_core_begin, ev_core_end, aux_stream, stream, nullptr);
}

extern "C" int cti_tcnet_forward_sm(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                                    const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                                    const float* const* rank_g, const float* const* rank_b, const float* T_g, float* out,
                                    uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                                    int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                                    void* ev_core_end, void* aux_stream, void* stream, float* sm_partials, size_t sm_partials_bytes) {
    CTI_REQUIRE_PTR(sm_partials); CTI_REQUIRE_PTR(zero_mask);
    const size_t need = cti_tcnet_softmax_partials_bytes(B, V, Q, A, h, G, prec);
    CTI_REQUIRE(need != 0, CTI_E_UNSUPPORTED, "cti_tcnet_forward_sm: no softmax partials for prec=%d G=%d h=%d (cti_tcnet_softmax_partials_bytes is 0)", prec, G, h);
    CTI_REQUIRE(sm_partials_bytes >= need, CTI_E_WORKSPACE, "cti_tcnet_forward_sm: partials block %zu < %zu", sm_partials_bytes, need);
    return tcnet_forward_impl(v, q, a, tucker_wv, tucker_g, tucker_b, rank_wv, rank_g, rank_b, T_g, out, zero_mask, B, V, Q, A, v_dim, q_dim, a_dim, h, R,
                              G, act, prec, prepared, workspace, workspace_bytes, ev_core_begin, ev_core_end, aux_stream, stream, sm_partials);
}

static int tcnet_forward_impl(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                              const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                              const float* const* rank_g, const float* const* rank_b, const float* T_g, float* out,
                              uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                              int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                              void* ev_core_end, void* aux_stream, void* stream, float* sm_part, float* p_fused,
                              const float* v_tucked, int64_t ld_vt, int v_rep, int v16) {
    // v16 (round 5; cti_triattention_forward_vt16): `v` AND `v_tucked` are bf16 rows -- v is then read for the zero-row mask only (the hoisted projection replaces
    // its Tucker layer), v_tucked enters the rank nets' product as a planes operand with a zero lo plane
    CTI_REQUIRE(!v16 || (v_tucked != nullptr && zero_mask != nullptr), CTI_E_UNSUPPORTED, "cti_triattention_forward_vt16: bf16 inputs need the hoisted v projection and the mask output");
    CTI_REQUIRE_PTR(v); CTI_REQUIRE_PTR(q); CTI_REQUIRE_PTR(a); CTI_REQUIRE_PTR(tucker_wv); CTI_REQUIRE_PTR(tucker_g);
    CTI_REQUIRE_PTR(tucker_b); CTI_REQUIRE_PTR(rank_wv); CTI_REQUIRE_PTR(rank_g); CTI_REQUIRE_PTR(rank_b); CTI_REQUIRE_PTR(T_g);
    CTI_REQUIRE_PTR(out); CTI_REQUIRE_PTR(workspace);
    Dims d{B, V, Q, A, v_dim, q_dim, a_dim, h, R, G};
    int rc = check_dims(d); if (rc) return rc;
    CTI_REQUIRE(act == CTI_ACT_NONE || act == CTI_ACT_RELU, CTI_E_UNSUPPORTED, "cti_tcnet_forward: act=%d", act);
    CTI_REQUIRE(prec == CTI_PREC_F32 || prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16 || prec == CTI_PREC_F16F6, CTI_E_UNSUPPORTED, "cti_tcnet_forward: prec=%d", prec);
    CTI_REQUIRE(prec != CTI_PREC_F16F6 || h % 32 == 0, CTI_E_UNSUPPORTED, "cti_tcnet_forward: the f16f6 mode needs h %% 32 == 0 (h=%d)", h);
    for (int s = 0; s < 3; ++s)
        CTI_REQUIRE(tucker_wv[s] && tucker_g[s] && tucker_b[s] && rank_wv[s] && rank_g[s] && rank_b[s], CTI_E_NULL, "cti_tcnet_forward: weight pointer %d is NULL", s);
    Plan p = carve(d, prec, workspace);
    CTI_REQUIRE(workspace_bytes >= p.bytes, CTI_E_WORKSPACE, "cti_tcnet_forward: workspace %zu < %zu", workspace_bytes, p.bytes);
    if (prepared) {                                             // scales, T_eff and weight planes come from cti_tcnet_prepare
        Bump wp{static_cast<char*>(const_cast<void*>(prepared)), 0, 0};
        carve_prep(d, prec, wp, p);
    }
    hipStream_t st = as_stream(stream);
    const int hr = h / R;
    const float* x[3] = {v, q, a};
    const int in[3] = {v_dim, q_dim, a_dim};
    const int64_t rows[3] = {(int64_t)B * V, (int64_t)B * Q, (int64_t)B * A};
    const int relu = act == CTI_ACT_RELU;

    // the all-zero-object mask (one pass over v): in front of everything -- except on the few-answer path with an auxiliary stream, where only the fused core's softmax
    // reads it and the main stream's chain (the a side) is the SHORTER of the two: there it is launched behind that chain, beside the auxiliary stream's v / q sides
    // (round 6: ~10 us off the critical chain of a c3 forward)
    const bool mask_late = zero_mask && aux_stream && prec != CTI_PREC_F32 && small_a(d);
    // (experiments' switches, read once; they place the guard's kernels on the main stream)
    static const bool order1_env = [] { const char* e = getenv("CTI_F6_ORDER"); return e && e[0] == '1'; }();
    static const bool join_before_rank = [] { const char* e = getenv("CTI_F6_JOIN"); return e && e[0] == 'r'; }();
    static const bool guard_front_env = [] { const char* e = getenv("CTI_F6_GUARD_FRONT"); return e && e[0] == '1'; }();
    // the guard block's reset: on chain B's stream where every guard kernel runs there or behind an event of it (round 6: as the main stream's first launch it sat
    // ~10 us in front of the encoding pass of `a`); on the main stream otherwise
    const bool reset_on_b = p.guard && aux_stream && !order1_env && !join_before_rank && !guard_front_env && !guard_late();
    if (p.guard && !reset_on_b) { rc = guard_reset(p.guard, st); if (rc) return rc; }
    if (zero_mask && !mask_late) {
        rc = v16 ? cti_zero_row_mask_bf16(v, v_dim, zero_mask, rows[0], v_dim, stream) : cti_zero_row_mask(v, v_dim, zero_mask, rows[0], v_dim, stream);
        if (rc) return rc;
    }
    if (!prepared) { rc = run_prepare(d, prec, p, tucker_wv, tucker_g, rank_wv, rank_g, T_g, false, stream); if (rc) return rc; }
    const int64_t mrows_per_b = (int64_t)V * Q * G;

    if (prec == CTI_PREC_F32) {
        for (int s = 0; s < 3; ++s) {
            rc = cti_wn_linear_fwd(x[s], in[s], tucker_wv[s], in[s], p.scale_t[s], h, tucker_b[s], p.t32[s], h, rows[s], in[s], h, act,
                                   CTI_PREC_F32, nullptr, 0, stream); if (rc) return rc;
            rc = cti_wn_linear_fwd(p.t32[s], h, rank_wv[s], h, p.scale_r[s], hr, rank_b[s], p.r32[s], h, rows[s], h, h, act,
                                   CTI_PREC_F32, nullptr, 0, stream); if (rc) return rc;
        }
        rc = mbuild_fast(p.r32[0], p.r32[1], p.Teff, p.M32, nullptr, nullptr, B, V, Q, R, hr, G, h, st);
        if (rc == CTI_E_UNSUPPORTED) rc = cti_paralind_mbuild_fwd(p.r32[0], p.r32[1], p.Teff, p.M32, B, V, Q, R, hr, hr, hr, G, stream);
        if (rc) return rc;
        if (ev_core_begin) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_begin), st);
        rc = cti_paralind_core_fwd(p.M32, p.r32[2], out, B, V * Q, A, G, h, CTI_PREC_F32, nullptr, 0, stream);
        if (ev_core_end) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_end), st);
        return rc;
    }

    const bool f6 = prec == CTI_PREC_F16F6;
    const bool fused_core = small_a(d);                        // A <= 6: mbuild_core_small replaces M build + planes + mode-3 GEMM
    CTI_REQUIRE(v_tucked == nullptr || (fused_core && v_rep >= 1 && B % v_rep == 0 && ld_vt >= h && h % 4 == 0 && (ld_vt & (v16 ? 7 : 3)) == 0 && (!v16 || h % 8 == 0) &&
                                        (reinterpret_cast<uintptr_t>(v_tucked) & 15) == 0),
                CTI_E_UNSUPPORTED, "cti_triattention_forward: a hoisted v projection needs the few-answer path, B %% v_rep == 0 and 16-B aligned rows (A=%d v_rep=%d)", A, v_rep);
    CTI_REQUIRE(!(fused_core && sm_part), CTI_E_UNSUPPORTED, "cti_tcnet_forward_sm: no softmax partials on the few-answer path (A=%d)", A);
    const int terms = prec == CTI_PREC_BF16 ? 1 : 3;
    // Two independent chains feed the mode-3 GEMM: chain A (the a side: split, Tucker, rank nets -- 2.8 ms at config 2, opens with the
    // HBM-bound split of `a`, which uses no LDS) and chain B (v and q sides + M build: 0.75 ms, LDS-heavy and latency-bound).  With
    // an auxiliary stream from the caller chain B runs beside chain A's split pass: fork/join with two events, no host sync.
    hipStream_t sb = aux_stream ? as_stream(aux_stream) : st;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_at = nullptr, ev_guard = nullptr;   // ev_at: the a side has encoded a~ (early scan) / A^ (the guard beside the mode-3 product); ev_guard: the verdict
    if (aux_stream) {
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess ||
            (p.guard && (hipEventCreateWithFlags(&ev_at, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_guard, hipEventDisableTiming) != hipSuccess)))
            return fail(CTI_E_UNSUPPORTED, "cti_tcnet_forward: hipEventCreate failed");
        (void)hipEventRecord(ev_fork, st);                  // scales, T_eff (and the mask) precede both chains
        (void)hipStreamWaitEvent(sb, ev_fork, 0);
    }
    if (reset_on_b) { rc = guard_reset(p.guard, sb); if (rc) { (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join); (void)hipEventDestroy(ev_at); (void)hipEventDestroy(ev_guard); return rc; } }
    auto finish = [&](int code) {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_at) (void)hipEventDestroy(ev_at);
        if (ev_guard) (void)hipEventDestroy(ev_guard);
        return code;
    };
    const int Kh = planes_kp(h);
    // v_tucked (cti_triattention_forward, few-answer path only): v's Tucker projection was computed by the caller -- hoisted into the batched GEMM of
    // the glimpses' pooling networks and, with v_rep > 1, once per IMAGE instead of once per row (the MC pipeline repeats every image per candidate
    // answer) -- as fp32 rows (B / v_rep * V, h) with row stride ld_vt; the rank nets then run on those rows (fp32 A operand split at
    // fragment-read time) and V^ holds one block per image.
    const bool hoisted_v = v_tucked != nullptr;
    int a_phase = 0;                                           // f16f6 a side: 0 = all of it, 1 = up to the Tucker product (+ ev_at), 2 = the rank nets' product
    auto side = [&](int s, hipStream_t ss) -> int {
        if (s == 0 && hoisted_v) {
            PlaneGemmArgs r{};
            r.Bh = p.wr[0].hi; r.Bl = p.wr[0].lo;
            if (v16) {                                           // bf16 rows: the hi plane as it stands, a zero lo plane for the three-product form
                // (round 6) plain bf16: cti_gemm16.hip reads the bf16 rows themselves where it takes the product (the small tiles are its own now): no split launch
                bool rows_direct = false;
                if (terms == 1) {
                    PlaneGemmArgs t = r;
                    t.Abf = v_tucked; t.ldabf = ld_vt; t.rows_allocA = rows[0] / v_rep; t.rows_allocB = p.wr[0].rows_alloc; t.nb1 = 1; t.nb2 = 1;
                    t.M = (int)(rows[0] / v_rep); t.N = h; t.Kp = planes_kp(h); t.terms = 1; t.scale = p.scale_r[0]; t.scale_div = hr; t.bias = rank_b[0]; t.relu = relu;
                    t.epi = 0; t.C = p.Vr; t.ldc_m = h; t.ldc_n = 1;
                    rows_direct = planes_kp(h) == h && gemm16_eligible(t);
                }
                if (rows_direct) { r.Abf = v_tucked; r.ldabf = ld_vt; r.rows_allocA = rows[0] / v_rep; }
                else {
                    int r1 = split_planes16(reinterpret_cast<const unsigned short*>(v_tucked), ld_vt, rows[0] / v_rep, h, p.tp[0].hi, terms == 3 ? p.tp[0].lo : nullptr, p.tp[0].rows_alloc, ss); if (r1) return r1;
                    r.Ah = p.tp[0].hi; r.Al = p.tp[0].lo; r.rows_allocA = p.tp[0].rows_alloc;
                }
            } else if (terms == 3) { r.Af = v_tucked; r.ldaf = ld_vt; r.Kreal = h; r.rows_allocA = rows[0] / v_rep + PLANE_SLACK_ROWS; }   // (the fp32-A operand path exists for the 3-term mode)
            else {
                int r1 = split_planes(v_tucked, ld_vt, rows[0] / v_rep, h, p.tp[0].hi, p.tp[0].lo, p.tp[0].rows_alloc, ss); if (r1) return r1;
                r.Ah = p.tp[0].hi; r.Al = p.tp[0].lo; r.rows_allocA = p.tp[0].rows_alloc;
            }
            r.rows_allocB = p.wr[0].rows_alloc; r.nb1 = 1; r.nb2 = 1;
            r.M = (int)(rows[0] / v_rep); r.N = h; r.Kp = planes_kp(h); r.terms = terms;
            r.scale = p.scale_r[0]; r.scale_div = hr; r.bias = rank_b[0]; r.relu = relu;
            r.epi = 0; r.C = p.Vr; r.ldc_m = h; r.ldc_n = 1;
            if (!prepared) { int r0 = split_planes(rank_wv[0], h, h, h, p.wr[0].hi, p.wr[0].lo, p.wr[0].rows_alloc, ss); if (r0) return r0; }
            return gemm_nt_planes(r, ss);
        }
        const bool f6_side = s == 2 && f6 && !fused_core;       // a side of the f16f6 mode: encode a -> transposed f16f6 Tucker product -> planes -> transposed f16f6 rank product -> planes
        const bool af32 = !f6_side && af32_side(prec, in[s]);
        if (af32 && (reinterpret_cast<uintptr_t>(x[s]) & 15)) return fail(CTI_E_ALIGN, "cti_tcnet_forward: input %d must be 16-B aligned (its rows are DMA'd as fp32)", s);
        int r_;
        if (!af32 && !f6_side) { r_ = split_planes(x[s], in[s], rows[s], in[s], p.xin[s].hi, p.xin[s].lo, p.xin[s].rows_alloc, ss); if (r_) return r_; }
        if (!prepared && !(f6_side && a_phase == 2)) {       // per-call weights: split beside this side's input (prepared: done once)
            r_ = split_planes(tucker_wv[s], in[s], h, in[s], p.wt[s].hi, p.wt[s].lo, p.wt[s].rows_alloc, ss); if (r_) return r_;
            r_ = split_planes(rank_wv[s], h, h, h, p.wr[s].hi, p.wr[s].lo, p.wr[s].rows_alloc, ss); if (r_) return r_;
            if (s == 2 && f6) { r_ = quantize_rank_a(d, p, rank_wv[2], tucker_wv[2], ss); if (r_) return r_; }
        }
        if (f6_side && a_phase == 2) {
            F6GemmArgs t{};
            t.A = p.f_wra; t.B = p.f_At; t.nb = 1; t.M = h; t.N = (int)rows[2]; t.epi = 6; t.out = &p.f_Arp; t.bias = rank_b[2]; t.relu = relu;
            return gemm_nt_f16f6(t, ss);
        }
        if (f6_side) {
            r_ = quantize_f16f6(x[2], in[2], rows[2], in[2], p.f_Ain, ss); if (r_) return r_;
            // (Measured and dropped: joining chain B HERE, behind the encoding pass, instead of in front of the mode-3 product.  Its kernels then run
            // unstarved -- the M build 0.36 ms of wall time instead of 1.2 -- but the step takes 4.72 instead of 4.60 ms: starved as they are, they
            // fill the tails of the persistent GEMMs' last tile rounds.)
            F6GemmArgs t{};                                  // a~^T = W_t a^T, then A^^T = W_r a~^T: rows = features, columns = the B * A answer tokens
            t.A = p.f_wta; t.B = p.f_Ain; t.nb = 1; t.M = h; t.N = (int)rows[2]; t.epi = 6; t.out = &p.f_At; t.bias = tucker_b[2]; t.relu = relu;
            r_ = gemm_nt_f16f6(t, ss); if (r_) return r_;
            if (ev_at) (void)hipEventRecord(ev_at, ss);
            if (a_phase == 1) return CTI_OK;
            t.A = p.f_wra; t.B = p.f_At; t.out = &p.f_Arp; t.bias = rank_b[2];
            return gemm_nt_f16f6(t, ss);
        }
        PlaneGemmArgs g{};                                   // Tucker: planes -> planes
        g.Ah = p.xin[s].hi; g.Al = p.xin[s].lo; g.Bh = p.wt[s].hi; g.Bl = p.wt[s].lo;
        if (af32) { g.Af = x[s]; g.ldaf = in[s]; g.Kreal = in[s]; }
        g.rows_allocA = p.xin[s].rows_alloc; g.rows_allocB = p.wt[s].rows_alloc; g.nb1 = 1; g.nb2 = 1;
        g.M = (int)rows[s]; g.N = h; g.Kp = p.xin[s].Kp; g.terms = terms; g.epi = 1;
        g.Ph = p.tp[s].hi; g.Pl = p.tp[s].lo; g.rows_allocP = p.tp[s].rows_alloc; g.Np = Kh;

        g.scale = p.scale_t[s]; g.scale_div = h; g.bias = tucker_b[s]; g.relu = relu;
        r_ = gemm_nt_planes(g, ss); if (r_) return r_;
        PlaneGemmArgs r{};                                   // packed rank nets: planes -> fp32 (v, q) or planes (a)
        r.Ah = p.tp[s].hi; r.Al = p.tp[s].lo; r.Bh = p.wr[s].hi; r.Bl = p.wr[s].lo;
        r.rows_allocA = p.tp[s].rows_alloc; r.rows_allocB = p.wr[s].rows_alloc; r.nb1 = 1; r.nb2 = 1;
        r.M = (int)rows[s]; r.N = h; r.Kp = Kh; r.terms = terms;
        r.scale = p.scale_r[s]; r.scale_div = hr; r.bias = rank_b[s]; r.relu = relu;
        if (s < 2) { r.epi = 0; r.C = s == 0 ? p.Vr : p.Qr; r.ldc_m = h; r.ldc_n = 1; }
        else if (fused_core) { r.epi = 0; r.C = p.Ar32; r.ldc_m = h; r.ldc_n = 1; }
        else if (f6) { r.epi = 4; r.f6out = &p.f_Arp; r.Np = h; }            // A^ straight into the f16 + fp6 planes of the mode-3 product
        else       { r.epi = 1; r.Ph = p.Arp.hi; r.Pl = p.Arp.lo; r.rows_allocP = p.Arp.rows_alloc; r.Np = Kh; }
        return gemm_nt_planes(r, ss);
    };
    // CTI_F6_ORDER=1 (round 6 experiment; f16f6 with an auxiliary stream and the direct-encoding M build): the v / q sides run FIRST, alone, on the main stream
    // (~0.1 ms), then the M build -- MFMA / vector-ALU bound, 256 workgroups -- runs on the auxiliary stream BESIDE the encoding pass of `a` (HBM bound) instead of
    // behind the a side's last product, where nothing overlaps it (0.27 ms + the scans in front of the mode-3 product: profiles/r05_step_timeline.txt)
    const bool order1 = order1_env && f6 && aux_stream && !fused_core && !guard_late() && !p.Mf32;
    // chain B on the auxiliary stream (or first, on the main stream)
    rc = side(0, order1 ? st : sb); if (rc) return finish(rc);
    rc = side(1, order1 ? st : sb); if (rc) return finish(rc);
    if (order1) {
        hipEvent_t ev_vq = nullptr;
        if (hipEventCreateWithFlags(&ev_vq, hipEventDisableTiming) != hipSuccess) return finish(fail(CTI_E_UNSUPPORTED, "cti_tcnet_forward: hipEventCreate failed"));
        (void)hipEventRecord(ev_vq, st);
        (void)hipStreamWaitEvent(sb, ev_vq, 0);
        (void)hipEventDestroy(ev_vq);
        {   // auxiliary stream: the fp32 sweeps of V^ / Q^ / T_eff, the M build
            GuardArgs gb{};
            gb.words = p.guard;
            gb.seg[gb.nseg++] = guard_seg_f32(p.Vr, rows[0] * h, 6);
            gb.seg[gb.nseg++] = guard_seg_f32(p.Qr, rows[1] * h, 7);
            gb.seg[gb.nseg++] = guard_seg_f32(p.Tt, (int64_t)R * hr * hr * hr * G, 8);
            if (!guard_ablate()) { rc = guard_scan(gb, sb); if (rc) return finish(rc); }
            rc = mbuild_mfma_f6(p.Vr, p.Qr, p.Tt, p.f_Mp, B, V, Q, R, hr, G, sb, p.Tj); if (rc) return finish(rc);
            (void)hipEventRecord(ev_join, sb);
        }
        // main stream: the whole a side, its scans, the join, the cancellation estimate + verdict, the mode-3 product
        a_phase = 0;
        rc = side(2, st); if (rc) return finish(rc);
        GuardArgs ga{};
        ga.words = p.guard; ga.final = 1; ga.n_slots = 9; ga.f32_slots = 7u << 6;
        ga.seg[ga.nseg++] = guard_seg_planes(p.f_wta, h, 1);
        ga.seg[ga.nseg++] = guard_seg_planes(p.f_wra, h, 2);
        ga.seg[ga.nseg++] = guard_seg_planes(p.f_Ain, rows[2], 3);
        ga.seg[ga.nseg++] = guard_seg_planes(p.f_At, rows[2], 4);
        ga.seg[ga.nseg++] = guard_seg_planes(p.f_Arp, rows[2], 5);
        (void)hipStreamWaitEvent(st, ev_join, 0);
        ga.seg[ga.nseg++] = guard_seg_planes(p.f_Mp, (int64_t)B * mrows_per_b, 0);
        if (!guard_ablate()) {
            rc = guard_cancel(p.f_Mp, mrows_per_b, p.f_Arp, A, B, p.guard, st); if (rc) return finish(rc);
            rc = guard_scan(ga, st); if (rc) return finish(rc);
            (void)hipEventRecord(ev_guard, st); (void)hipStreamWaitEvent(sb, ev_guard, 0);      // "an event recorded on aux_stream after the call marks the verdict" (cti_hip.h)
        }
        F6GemmArgs c{};
        c.A = p.f_Mp; c.B = p.f_Arp; c.rA = p.f_Mp.rstride; c.rB = p.f_Arp.rstride; c.nb = B; c.M = (int)mrows_per_b; c.N = A;
        c.epi = 3; c.gdiv = G; c.C = out; c.ldc_m = (int64_t)A * G; c.ldc_n = G; c.sC = (int64_t)V * Q * A * G;
        if (sm_part) { c.sm_part = sm_part; c.sm_mask = zero_mask; c.sm_rows_per_obj = Q * G; c.sm_objs = V; }
        if (ev_core_begin) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_begin), st);
        rc = gemm_nt_f16f6(c, st);
        if (ev_core_end) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_end), st);
        if (rc) return finish(rc);
        return finish(guard_ablate() ? CTI_OK : guard_poison(p.guard, out, (int64_t)B * V * Q * A * G, st));
    }
    if (fused_core) {
        // few answer tokens: chain B ends with the rank nets; modes 1 + 2 + 3 are one kernel behind the join, M is never written
        if (aux_stream) (void)hipEventRecord(ev_join, sb);
        rc = side(2, st); if (rc) return finish(rc);
        if (mask_late) {
            rc = v16 ? cti_zero_row_mask_bf16(v, v_dim, zero_mask, rows[0], v_dim, stream) : cti_zero_row_mask(v, v_dim, zero_mask, rows[0], v_dim, stream);
            if (rc) return finish(rc);
        }
        if (aux_stream) (void)hipStreamWaitEvent(st, ev_join, 0);
        if (ev_core_begin) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_begin), st);
        rc = mbuild_core_small(p.Vr, p.Qr, p.Tt, p.Ar32, out, B, V, Q, A, R, hr, G, st, p_fused ? zero_mask : nullptr, p_fused, hoisted_v ? v_rep : 1, terms, p.Tk);   // p_fused: + the masked softmax
        if (ev_core_end) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_end), st);
        return finish(rc);
    }
    // Range guard (cti_f16f6_guard.hip).  Three scans keep it off the critical path, which is chain B's END (the M build, starved of CUs by the
    // persistent a-side GEMMs, finishes ~0.3 ms after them: profiles/r03 timeline):
    //   early  (auxiliary stream, BEFORE the M build, behind the a side's Tucker product): the scale bytes of `a`, a~ and the a-side weights + a
    //          non-finite sweep of V^ / Q^ / T_eff;
    //   middle (main stream, right behind the rank nets' product, while the main stream would otherwise just wait for chain B): A^;
    //   final  (main stream, behind the join): M's 8 MB of scale bytes, then the verdict -- ~10 us in front of the mode-3 product.
    // Without an auxiliary stream everything is in stream order anyway: early = the fp32 sweeps, final = the a-side weights (per-call weights are
    // encoded inside side(2), AFTER the early scan: scanning them early read uninitialised workspace) + the four encoded tensors.
    auto early_scan = [&](bool with_a) -> int {
        GuardArgs gb{};
        gb.words = p.guard;
        if (with_a) {                                            // (no auxiliary stream: per-call weights are encoded inside side(2) BELOW -- the final scan takes them)
            gb.seg[gb.nseg++] = guard_seg_planes(p.f_wta, h, 1);
            gb.seg[gb.nseg++] = guard_seg_planes(p.f_wra, h, 2);
        }
        if (with_a) {
            gb.seg[gb.nseg++] = guard_seg_planes(p.f_Ain, rows[2], 3);
            gb.seg[gb.nseg++] = guard_seg_planes(p.f_At, rows[2], 4);
        }
        gb.seg[gb.nseg++] = guard_seg_f32(p.Vr, rows[0] * h, 6);
        gb.seg[gb.nseg++] = guard_seg_f32(p.Qr, rows[1] * h, 7);
        gb.seg[gb.nseg++] = guard_seg_f32(p.Tt, (int64_t)R * hr * hr * hr * G, 8);
        if (guard_ablate()) return CTI_OK;
        return guard_scan(gb, sb);
    };
    // Round 6: with an auxiliary stream the guard's kernels leave the critical path altogether.  They need 35-39 registers and <= 544 B of LDS, and the mode-3
    // product's persistent workgroups leave every SIMD 96 registers and every CU 1.5 KiB: ONE scan of everything (`a`, a~, A^, M, the a-side weights, the
    // fp32 sweeps) + the cancellation estimate + the verdict run on the auxiliary stream BESIDE the mode-3 product, behind the M build and an event of the
    // rank nets' product; the main stream meets them again in front of the NaN fill.  (Rounds 3-5: early / middle / final scans in front of the product,
    // 0.10-0.15 ms of the step by the guard-off A/B, profiles/r05_guard_ab.txt; CTI_F6_GUARD_FRONT=1 restores that placement for the A/B.)
    const bool early_join = f6 && aux_stream && join_before_rank;
    const bool guard_beside = f6 && aux_stream && p.guard && !guard_front_env && !guard_late() && !early_join;
    if (f6 && aux_stream) {
        // host order matters: the main stream's first half (encode `a`, Tucker product, ev_at) is enqueued BEFORE chain B waits for ev_at
        a_phase = 1;
        rc = side(2, st); if (rc) return finish(rc);
        if (!guard_late() && !guard_beside) {
            (void)hipStreamWaitEvent(sb, ev_at, 0);
            rc = early_scan(true); if (rc) return finish(rc);
        }
    } else if (f6 && !guard_late()) {
        rc = early_scan(false); if (rc) return finish(rc);
    }
    if (f6) {
        // M as fp32 rows (MFMA M build, or the VALU forms for other shapes), then one encoding pass into planes whose batches of V*Q*G rows
        // start at multiples of 8 rows
        rc = p.Mf32 ? CTI_E_UNSUPPORTED : mbuild_mfma_f6(p.Vr, p.Qr, p.Tt, p.f_Mp, B, V, Q, R, hr, G, sb, p.Tj);       // the planes directly (hr = 16, glimpse 2, X + hold buffer fit the LDS)
        if (rc == CTI_E_UNSUPPORTED && !p.Mf32) return finish(fail(CTI_E_UNSUPPORTED, "cti_tcnet_forward: the direct-encoding M build refused a shape its plan accepted"));
        if (rc == CTI_E_UNSUPPORTED) {
            rc = mbuild_mfma(p.Vr, p.Qr, p.Tt, nullptr, nullptr, p.Mf32, B, V, Q, R, hr, G, h, sb);
            if (rc == CTI_E_UNSUPPORTED) rc = mbuild_fast(p.Vr, p.Qr, p.Teff, p.Mf32, nullptr, nullptr, B, V, Q, R, hr, G, h, sb);
            if (rc == CTI_E_UNSUPPORTED) rc = cti_paralind_mbuild_fwd(p.Vr, p.Qr, p.Teff, p.Mf32, B, V, Q, R, hr, hr, hr, G, sb);
            if (rc) return finish(rc);
            rc = quantize_f16f6(p.Mf32, h, (int64_t)B * mrows_per_b, h, p.f_Mp, sb);
        }
    } else {
    rc = mbuild_mfma(p.Vr, p.Qr, p.Tt, p.Mp.hi, p.Mp.lo, nullptr, B, V, Q, R, hr, G, p.Mp.rows_alloc * 16, sb);
    if (rc == CTI_E_UNSUPPORTED) rc = mbuild_fast(p.Vr, p.Qr, p.Teff, nullptr, p.Mp.hi, p.Mp.lo, B, V, Q, R, hr, G, p.Mp.rows_alloc * 16, sb);
    if (rc == CTI_E_UNSUPPORTED) {
        // generic M build writes fp32 (B,V,Q,G,h): borrow `out` as scratch when it is large enough (B*V*Q*A*G >= B*V*Q*G*h), else the plan
        // carved a scratch block (m_needs_scratch)
        float* mscr = p.Mscr ? p.Mscr : out;
        if (!p.Mscr && A < h) return finish(fail(CTI_E_UNSUPPORTED, "cti_tcnet_forward: the plan carved no M scratch for a shape its plane M builds refuse (V=%d Q=%d h/rank=%d A=%d)", V, Q, hr, A));
        rc = cti_paralind_mbuild_fwd(p.Vr, p.Qr, p.Teff, mscr, B, V, Q, R, hr, hr, hr, G, sb); if (rc) return finish(rc);
        rc = split_planes(mscr, h, (int64_t)B * mrows_per_b, h, p.Mp.hi, p.Mp.lo, p.Mp.rows_alloc, sb);
    }
    }
    if (rc) return finish(rc);
    if (aux_stream) (void)hipEventRecord(ev_join, sb);
    // chain A on the main stream (f16f6 with an auxiliary stream: its second half -- the first was enqueued above)
    a_phase = (f6 && aux_stream) ? 2 : 0;
    // Where the main stream joins chain B (f16f6, auxiliary stream).  The M build cannot share a CU with the persistent a-side GEMMs (153 KB and
    // 160 KB of LDS), so its 0.34 ms are exclusive wherever they fall: launched beside the rank nets' product it crawls in that product's tail and
    // finishes ~0.3 ms AFTER it (timeline in profiles/r03_step_timeline.txt) with the mode-3 product waiting.  CTI_F6_JOIN=rank (experiment):
    // join BEFORE the rank nets' product instead -- the M build then runs alone between the two a-side products.
    if (early_join) (void)hipStreamWaitEvent(st, ev_join, 0);
    rc = side(2, st); if (rc) return finish(rc);
    if (guard_beside) (void)hipEventRecord(ev_at, st);     // A^ is encoded: what the auxiliary stream's guard kernels wait for
    if (f6 && aux_stream && !early_join && !guard_late() && !guard_beside) {  // middle scan: A^, while the main stream would otherwise only wait for chain B
        GuardArgs gm{};
        gm.words = p.guard;
        gm.seg[gm.nseg++] = guard_seg_planes(p.f_Arp, rows[2], 5);
        if (!guard_ablate()) { rc = guard_scan(gm, st); if (rc) return finish(rc); }
    }
    if (aux_stream && !early_join) (void)hipStreamWaitEvent(st, ev_join, 0);
    if (f6) {
        F6GemmArgs c{};                                      // mode 3 + rank sum on the f16 + fp6 planes
        c.A = p.f_Mp; c.B = p.f_Arp; c.rA = p.f_Mp.rstride; c.rB = p.f_Arp.rstride; c.nb = B; c.M = (int)mrows_per_b; c.N = A;
        c.epi = 3; c.gdiv = G; c.C = out; c.ldc_m = (int64_t)A * G; c.ldc_n = G; c.sC = (int64_t)V * Q * A * G;
        if (sm_part) { c.sm_part = sm_part; c.sm_mask = zero_mask; c.sm_rows_per_obj = Q * G; c.sm_objs = V; }     // the Tri softmax's partial pass, from the accumulators
        // Range guard, final scan: M (8 MB of scale bytes at configs[1]; without an auxiliary stream `a`, a~ and A^ as well), then the verdict.
        // It leaves the status word BEFORE ev_core_begin, so a host that waits for that event learns it while the mode-3 product is still
        // running; the NaN fill behind the product needs no host at all.
        GuardArgs ga{};
        ga.words = p.guard; ga.final = 1; ga.n_slots = 9; ga.f32_slots = 7u << 6;
        ga.seg[ga.nseg++] = guard_seg_planes(p.f_Mp, (int64_t)B * mrows_per_b, 0);
        if (!aux_stream || guard_late() || guard_beside) {
            ga.seg[ga.nseg++] = guard_seg_planes(p.f_wta, h, 1);             // written by side(2) above when the caller keeps no prepared block
            ga.seg[ga.nseg++] = guard_seg_planes(p.f_wra, h, 2);
            ga.seg[ga.nseg++] = guard_seg_planes(p.f_Ain, rows[2], 3);
            ga.seg[ga.nseg++] = guard_seg_planes(p.f_At, rows[2], 4);
        }
        if (!aux_stream || early_join || guard_late() || guard_beside) ga.seg[ga.nseg++] = guard_seg_planes(p.f_Arp, rows[2], 5);
        if (guard_late() || guard_beside) {                  // ONE scan for everything
            ga.seg[ga.nseg++] = guard_seg_f32(p.Vr, rows[0] * h, 6);
            ga.seg[ga.nseg++] = guard_seg_f32(p.Qr, rows[1] * h, 7);
            ga.seg[ga.nseg++] = guard_seg_f32(p.Tt, (int64_t)R * hr * hr * hr * G, 8);
        }
        const bool guard_on = p.guard && !guard_ablate();
        if (guard_on && guard_beside) {                      // auxiliary stream, behind the M build (stream order) and A^ (ev_at): runs while the mode-3 product does
            (void)hipStreamWaitEvent(sb, ev_at, 0);
            rc = guard_cancel(p.f_Mp, mrows_per_b, p.f_Arp, A, B, p.guard, sb); if (rc) return finish(rc);
            rc = guard_scan(ga, sb); if (rc) return finish(rc);
            (void)hipEventRecord(ev_guard, sb);
        } else if (guard_on) {
            rc = guard_cancel(p.f_Mp, mrows_per_b, p.f_Arp, A, B, p.guard, st); if (rc) return finish(rc);   // the estimate the final scan's verdict reads
            rc = guard_scan(ga, st); if (rc) return finish(rc);
            if (aux_stream) { (void)hipEventRecord(ev_guard, st); (void)hipStreamWaitEvent(sb, ev_guard, 0); }   // the verdict is visible on the auxiliary stream too (cti_hip.h)
        }
        if (ev_core_begin) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_begin), st);
        rc = gemm_nt_f16f6(c, st);
        if (ev_core_end) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_end), st);
        if (rc) return finish(rc);
        if (guard_on && guard_beside) (void)hipStreamWaitEvent(st, ev_guard, 0);
        return finish(guard_ablate() ? CTI_OK : guard_poison(p.guard, out, (int64_t)B * V * Q * A * G, st));
    }
    PlaneGemmArgs c{};                                       // mode 3 + rank sum: rows (vq,g) x columns a, per sample
    c.Ah = p.Mp.hi; c.Al = p.Mp.lo; c.Bh = p.Arp.hi; c.Bl = p.Arp.lo;
    c.rows_allocA = p.Mp.rows_alloc; c.rows_allocB = p.Arp.rows_alloc; c.rA1 = mrows_per_b; c.rB1 = A; c.nb1 = B; c.nb2 = 1;
    c.M = (int)mrows_per_b; c.N = A; c.Kp = Kh; c.terms = terms; c.epi = 3; c.gdiv = G;
    c.C = out; c.ldc_m = (int64_t)A * G; c.ldc_n = G; c.sC1 = (int64_t)V * Q * A * G;
    if (ev_core_begin) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_begin), st);
    rc = gemm_nt_planes(c, st);
    if (ev_core_end) (void)hipEventRecord(static_cast<hipEvent_t>(ev_core_end), st);
    return finish(rc);
}


// ---- TriAttention.forward (reference src/attention.py:49-59) as ONE call: logits = TCNet.forward, -inf on the all-zero rows of v, softmax over
// the flattened (v, q, a) axis per glimpse.  Few answer tokens (the FFOE / MC models): the fused modes-1+2+3 kernel holds a sample's logits in
// registers and writes `logits` and `p` itself -- no softmax launches at all.  f16f6 mode with glimpse 2: the mode-3 product leaves the softmax's
// partial pass, one normalise pass follows.  Otherwise: the two-pass masked softmax.
static size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

/* 1 when cti_triattention_forward accepts v_tucker_out for this shape / mode (the fused few-answer path), else 0 */
extern "C" int cti_triattention_hoist_ok(int B, int V, int Q, int A, int h, int R, int G, int prec) {
    if (B <= 0 || V <= 0 || Q <= 0 || A <= 0 || h <= 0 || R <= 0 || G <= 0 || h % R || h % 4) return 0;
    if (!(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16 || (prec == CTI_PREC_F16F6 && h % 32 == 0))) return 0;
    Dims d{B, V, Q, A, 1, 1, 1, h, R, G};
    return small_a(d) ? 1 : 0;
}

extern "C" size_t cti_triattention_workspace_bytes(int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R, int G, int prec) {
    const size_t t = cti_tcnet_forward_workspace_bytes(B, V, Q, A, v_dim, q_dim, a_dim, h, R, G, prec);
    if (t == 0) return 0;
    return a256(t) + a256(cti_tcnet_softmax_partials_bytes(B, V, Q, A, h, G, prec)) + a256(cti_softmax_tri_workspace_bytes(B, V, (int64_t)Q * A, G));
}

static int triattention_impl(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                             const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                             const float* const* rank_g, const float* const* rank_b, const float* T_g, float* logits, float* p_out,
                             uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                             int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                             void* ev_core_end, void* aux_stream, void* stream, const float* v_tucker_out, int64_t ld_vt, int v_rep, int v16);

extern "C" int cti_triattention_forward(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                                        const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                                        const float* const* rank_g, const float* const* rank_b, const float* T_g, float* logits, float* p_out,
                                        uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                                        int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                                        void* ev_core_end, void* aux_stream, void* stream, const float* v_tucker_out, int64_t ld_vt, int v_rep) {
    return triattention_impl(v, q, a, tucker_wv, tucker_g, tucker_b, rank_wv, rank_g, rank_b, T_g, logits, p_out, zero_mask, B, V, Q, A, v_dim, q_dim, a_dim, h, R, G, act, prec,
                             prepared, workspace, workspace_bytes, ev_core_begin, ev_core_end, aux_stream, stream, v_tucker_out, ld_vt, v_rep, 0);
}

// Round 5 (BASELINE configs[2] / [3] name bf16 tensors): the same call with `v` (B, V, v_dim) AND the hoisted projection `v_tucker_out` as bf16 rows (ld_vt in
// elements, a multiple of 8).  v is read for the zero-row mask only; the few-answer path (cti_triattention_hoist_ok) and a hoisted projection are required --
// CTI_E_UNSUPPORTED otherwise (the caller widens v and takes cti_triattention_forward).
extern "C" int cti_triattention_forward_vt16(const void* v_bf16, const float* q, const float* a, const float* const* tucker_wv,
                                             const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                                             const float* const* rank_g, const float* const* rank_b, const float* T_g, float* logits, float* p_out,
                                             uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                                             int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                                             void* ev_core_end, void* aux_stream, void* stream, const void* v_tucker_out_bf16, int64_t ld_vt, int v_rep) {
    CTI_REQUIRE_PTR(v_tucker_out_bf16);
    CTI_REQUIRE(cti_triattention_hoist_ok(B, V, Q, A, h, R, G, prec), CTI_E_UNSUPPORTED, "cti_triattention_forward_vt16: the few-answer path only (A=%d)", A);
    return triattention_impl(static_cast<const float*>(v_bf16), q, a, tucker_wv, tucker_g, tucker_b, rank_wv, rank_g, rank_b, T_g, logits, p_out, zero_mask, B, V, Q, A, v_dim, q_dim,
                             a_dim, h, R, G, act, prec, prepared, workspace, workspace_bytes, ev_core_begin, ev_core_end, aux_stream, stream,
                             static_cast<const float*>(v_tucker_out_bf16), ld_vt, v_rep, 1);
}

static int triattention_impl(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                             const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                             const float* const* rank_g, const float* const* rank_b, const float* T_g, float* logits, float* p_out,
                             uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                             int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                             void* ev_core_end, void* aux_stream, void* stream, const float* v_tucker_out, int64_t ld_vt, int v_rep, int v16) {
    CTI_REQUIRE_PTR(p_out); CTI_REQUIRE_PTR(zero_mask); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(G >= 2, CTI_E_UNSUPPORTED, "cti_triattention_forward: glimpse must be >= 2 (the reference's mask expand fails for 1, src/attention.py:55)");
    const size_t need = cti_triattention_workspace_bytes(B, V, Q, A, v_dim, q_dim, a_dim, h, R, G, prec);
    CTI_REQUIRE(need != 0, CTI_E_SHAPE, "cti_triattention_forward: B=%d V=%d Q=%d A=%d h=%d R=%d G=%d", B, V, Q, A, h, R, G);
    CTI_REQUIRE(workspace_bytes >= need, CTI_E_WORKSPACE, "cti_triattention_forward: workspace %zu < %zu", workspace_bytes, need);
    const size_t wt = cti_tcnet_forward_workspace_bytes(B, V, Q, A, v_dim, q_dim, a_dim, h, R, G, prec);
    const size_t pb = cti_tcnet_softmax_partials_bytes(B, V, Q, A, h, G, prec);
    const size_t sb = cti_softmax_tri_workspace_bytes(B, V, (int64_t)Q * A, G);
    char* w = static_cast<char*>(workspace);
    float* part = reinterpret_cast<float*>(w + a256(wt));
    void* sws = w + a256(wt) + a256(pb);
    Dims d{B, V, Q, A, v_dim, q_dim, a_dim, h, R, G};
    int rc = check_dims(d); if (rc) return rc;
    const bool planes_mode = prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16 || (prec == CTI_PREC_F16F6 && h % 32 == 0);
    if (planes_mode && small_a(d))                             // logits + p out of the fused kernel's registers
        return tcnet_forward_impl(v, q, a, tucker_wv, tucker_g, tucker_b, rank_wv, rank_g, rank_b, T_g, logits, zero_mask, B, V, Q, A, v_dim, q_dim, a_dim, h, R,
                                  G, act, prec, prepared, workspace, wt, ev_core_begin, ev_core_end, aux_stream, stream, nullptr, p_out, v_tucker_out, ld_vt, v_rep > 0 ? v_rep : 1, v16);
    CTI_REQUIRE(v_tucker_out == nullptr, CTI_E_UNSUPPORTED, "cti_triattention_forward: a hoisted v projection is taken on the few-answer path only (cti_triattention_hoist_ok)");
    const bool partials = pb != 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && ((int64_t)V * Q * A) % 2 == 0;
    rc = tcnet_forward_impl(v, q, a, tucker_wv, tucker_g, tucker_b, rank_wv, rank_g, rank_b, T_g, logits, zero_mask, B, V, Q, A, v_dim, q_dim, a_dim, h, R,
                            G, act, prec, prepared, workspace, wt, ev_core_begin, ev_core_end, aux_stream, stream, partials ? part : nullptr);
    if (rc) return rc;
    if (partials) return cti_masked_softmax_tri_from_partials_fwd(logits, zero_mask, part, pb, p_out, B, V, (int64_t)Q * A, G, sws, sb, stream);
    return cti_masked_softmax_tri_fwd(logits, zero_mask, p_out, B, V, (int64_t)Q * A, G, sws, sb, stream);
}
