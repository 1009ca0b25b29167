// cti_api.hip -- error reporting and the C-ABI entry points that dispatch to the MFMA GEMMs.
#include "cti_common.h"
#include <string.h>
#include <atomic>

namespace cti {

char* err_buf() {
    static thread_local char buf[512] = "no error";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// Tuning overrides are THREAD-LOCAL, like the error string: the header promises no process-wide mutable state, and a test thread that forces a
// tile geometry must not change what another thread's production call launches.
static thread_local int g_gemm_cfg = -1;
static thread_local long long g_tri_chunk = 0;
static thread_local int g_guard_rho_milli[2] = {2750, 5500};      // cancellation estimate beyond which a guarded f16f6 call is re-run as bf16x3 / as exact fp32 (x 1000)
static thread_local int g_guard_poison_bits = 31;                 // status bits that NaN-fill the output of a guarded call
static thread_local int g_guard_strata = 1;                       // test knob: 0 = the cancellation estimate samples 32 evenly spaced rows per operand (round 4)
static thread_local int g_gemm16_sk = -1;                          // stream-K cut of the plain-bf16 row GEMM: -1 = where it was measured to pay (three or more rounds of tiles), 0 = never, 1 = wherever it can be planned
static thread_local int g_gru_persistent = 0;                      // 1 = cti_gru_forward may run all steps of an eligible call as ONE persistent launch (the caller keeps two of them off the chip at once)
static thread_local int g_f6_core_free_cus = -1;                  // CUs the mode-3 product leaves to the guard kernels beside it (-1 = the library's default)
int tuning_gemm_cfg() { return g_gemm_cfg; }
int64_t tuning_tri_chunk() { return (int64_t)g_tri_chunk; }
float tuning_guard_rho(int which) { return 1e-3f * (float)g_guard_rho_milli[which ? 1 : 0]; }
unsigned tuning_guard_poison_bits() { return (unsigned)g_guard_poison_bits; }
int tuning_f6_core_free_cus() { return g_f6_core_free_cus; }
int tuning_guard_strata() { return g_guard_strata; }
int tuning_gemm16_sk() { return g_gemm16_sk; }
int tuning_gru_persistent() { return g_gru_persistent; }

}  // namespace cti

using namespace cti;

extern "C" int cti_set_tuning(int key, int64_t value) {
    switch (key) {
        case CTI_TUNE_GEMM_CFG:
            CTI_REQUIRE(value >= -1 && value <= 2, CTI_E_SHAPE, "cti_set_tuning: GEMM_CFG must be -1 (auto), 0, 1 or 2, got %lld", (long long)value);
            g_gemm_cfg = (int)value; return CTI_OK;
        case CTI_TUNE_TRI_CHUNK:
            CTI_REQUIRE(value == 0 || (value >= 4 && value <= (1ll << 30) && value % 4 == 0), CTI_E_SHAPE,
                        "cti_set_tuning: TRI_CHUNK must be 0 (auto) or a multiple of 4 in [4, 2^30], got %lld", (long long)value);
            g_tri_chunk = (long long)value; return CTI_OK;
        case CTI_TUNE_GUARD_RHO_BF16X3:
        case CTI_TUNE_GUARD_RHO_FP32:
            CTI_REQUIRE(value >= 1 && value <= 1000000000ll, CTI_E_SHAPE, "cti_set_tuning: a guard threshold is rho x 1000 in [1, 1e9], got %lld", (long long)value);
            g_guard_rho_milli[key == CTI_TUNE_GUARD_RHO_FP32] = (int)value; return CTI_OK;
        case CTI_TUNE_GUARD_POISON_BITS:
            CTI_REQUIRE(value >= 0 && value <= 31, CTI_E_SHAPE, "cti_set_tuning: GUARD_POISON_BITS is a mask of the five status bits, got %lld", (long long)value);
            g_guard_poison_bits = (int)value; return CTI_OK;
        case CTI_TUNE_GUARD_STRATA:
            CTI_REQUIRE(value == 0 || value == 1, CTI_E_SHAPE, "cti_set_tuning: GUARD_STRATA is 0 or 1, got %lld", (long long)value);
            g_guard_strata = (int)value; return CTI_OK;
        case CTI_TUNE_F6_CORE_FREE_CUS:
            CTI_REQUIRE(value >= -1 && value <= 64, CTI_E_SHAPE, "cti_set_tuning: F6_CORE_FREE_CUS must be -1 (default) or 0 .. 64, got %lld", (long long)value);
            g_f6_core_free_cus = (int)value; return CTI_OK;
        case CTI_TUNE_GEMM16_SK:
            CTI_REQUIRE(value >= -1 && value <= 1, CTI_E_SHAPE, "cti_set_tuning: GEMM16_SK must be -1 (auto), 0 or 1, got %lld", (long long)value);
            g_gemm16_sk = (int)value; return CTI_OK;
        case CTI_TUNE_GRU_PERSISTENT:
            CTI_REQUIRE(value >= 0 && value <= 2, CTI_E_SHAPE, "cti_set_tuning: GRU_PERSISTENT must be 0, 1 or 2, got %lld", (long long)value);
            g_gru_persistent = (int)value; return CTI_OK;
        default: return fail(CTI_E_UNSUPPORTED, "cti_set_tuning: unknown key %d", key);
    }
}
extern "C" int64_t cti_get_tuning(int key) {
    switch (key) {
        case CTI_TUNE_GEMM_CFG: return tuning_gemm_cfg();
        case CTI_TUNE_TRI_CHUNK: return tuning_tri_chunk();
        case CTI_TUNE_GUARD_RHO_BF16X3: return g_guard_rho_milli[0];
        case CTI_TUNE_GUARD_RHO_FP32: return g_guard_rho_milli[1];
        case CTI_TUNE_GUARD_POISON_BITS: return g_guard_poison_bits;
        case CTI_TUNE_F6_CORE_FREE_CUS: return g_f6_core_free_cus;
        case CTI_TUNE_GUARD_STRATA: return g_guard_strata;
        case CTI_TUNE_GEMM16_SK: return g_gemm16_sk;
        case CTI_TUNE_GRU_PERSISTENT: return g_gru_persistent;
        default: return INT64_MIN;
    }
}

extern "C" int cti_abi_version(void) { return CTI_ABI_VERSION; }
extern "C" const char* cti_last_error_string(void) { return err_buf(); }

extern "C" void* cti_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { fail(CTI_E_UNSUPPORTED, "cti_event_create: hipEventCreate failed"); return nullptr; }
    return e;
}
extern "C" int cti_event_destroy(void* ev) { return ev ? (int)hipEventDestroy(static_cast<hipEvent_t>(ev)) : CTI_OK; }
extern "C" int cti_event_record(void* ev, void* stream) {
    CTI_REQUIRE_PTR(ev);
    hipError_t e = hipEventRecord(static_cast<hipEvent_t>(ev), as_stream(stream));
    return e == hipSuccess ? CTI_OK : fail((int)e, "cti_event_record: %s", hipGetErrorString(e));
}
extern "C" int cti_event_elapsed_ms(void* begin, void* end, float* ms) {
    CTI_REQUIRE_PTR(begin); CTI_REQUIRE_PTR(end); CTI_REQUIRE_PTR(ms);
    hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(end));
    if (e == hipSuccess) e = hipEventElapsedTime(ms, static_cast<hipEvent_t>(begin), static_cast<hipEvent_t>(end));
    return e == hipSuccess ? CTI_OK : fail((int)e, "cti_event_elapsed_ms: %s", hipGetErrorString(e));
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t cti_wn_linear_workspace_bytes(int64_t rows, int in_dim, int out_dim, int prec) {
    if (prec == CTI_PREC_F32 || rows <= 0 || in_dim <= 0 || out_dim <= 0) return 0;
    size_t n = align256(planes_bytes(rows + PLANE_SLACK_ROWS, in_dim)) + align256(planes_bytes(out_dim + PLANE_SLACK_ROWS, in_dim));
    if (rows < (1ll << 31)) {                                               // split-K partials (skinny layers: rows = batch size)
        const int S = plan_ksplit((int)rows, out_dim, planes_kp(in_dim), 1);
        if (S > 1) n += align256(sizeof(float) * (size_t)S * (size_t)rows * (size_t)out_dim);
    }
    return n;
}

extern "C" int cti_wn_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* scale, int scale_div,
                                 const float* bias, float* y, int64_t ldy, int64_t rows, int in_dim, int out_dim, int act,
                                 int prec, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(w); CTI_REQUIRE_PTR(y);
    CTI_REQUIRE(rows > 0 && in_dim > 0 && out_dim > 0 && rows < (1ll << 31), CTI_E_SHAPE,
                "cti_wn_linear_fwd: rows=%lld in=%d out=%d", (long long)rows, in_dim, out_dim);
    CTI_REQUIRE(ldx >= in_dim && ldw >= in_dim && ldy >= out_dim, CTI_E_SHAPE, "cti_wn_linear_fwd: ldx=%lld ldw=%lld ldy=%lld",
                (long long)ldx, (long long)ldw, (long long)ldy);
    CTI_REQUIRE(act == CTI_ACT_NONE || act == CTI_ACT_RELU, CTI_E_UNSUPPORTED, "cti_wn_linear_fwd: act=%d", act);
    CTI_REQUIRE(scale == nullptr || scale_div > 0, CTI_E_SHAPE, "cti_wn_linear_fwd: scale_div=%d", scale_div);
    GemmP p{};
    p.A = x; p.B = w; p.C = y;
    p.lda = ldx; p.ldb = ldw; p.ldc_m = ldy; p.ldc_n = 1;
    p.nb1 = 1; p.nb2 = 1;
    p.M = (int)rows; p.N = out_dim; p.K = in_dim;
    p.scale = scale; p.scale_div = scale ? scale_div : 1; p.bias = bias; p.relu = (act == CTI_ACT_RELU);
    if (prec == CTI_PREC_F32) return gemm_nt_f32(p, as_stream(stream));
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_wn_linear_fwd: precision mode %d", prec);
    CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(workspace_bytes >= cti_wn_linear_workspace_bytes(rows, in_dim, out_dim, prec), CTI_E_WORKSPACE,
                "cti_wn_linear_fwd: workspace %zu < %zu", workspace_bytes, cti_wn_linear_workspace_bytes(rows, in_dim, out_dim, prec));
    {
        const int Kp = planes_kp(in_dim);
        const int64_t ra = rows + PLANE_SLACK_ROWS, rb = out_dim + PLANE_SLACK_ROWS;
        unsigned short* xh = static_cast<unsigned short*>(workspace);
        unsigned short* xl = xh + (size_t)ra * Kp;
        unsigned short* wh = reinterpret_cast<unsigned short*>(static_cast<char*>(workspace) + align256(planes_bytes(ra, in_dim)));
        unsigned short* wl = wh + (size_t)rb * Kp;
        int rc = split_planes(x, ldx, rows, in_dim, xh, xl, ra, as_stream(stream)); if (rc) return rc;
        rc = split_planes(w, ldw, out_dim, in_dim, wh, wl, rb, as_stream(stream)); if (rc) return rc;
        PlaneGemmArgs g{};
        g.Ah = xh; g.Al = xl; g.Bh = wh; g.Bl = wl; g.rows_allocA = ra; g.rows_allocB = rb; g.nb1 = 1; g.nb2 = 1;
        g.M = (int)rows; g.N = out_dim; g.Kp = Kp; g.terms = prec == CTI_PREC_BF16X3 ? 3 : 1; g.epi = 0;
        g.C = y; g.ldc_m = ldy; g.ldc_n = 1;
        g.scale = scale; g.scale_div = scale ? scale_div : 1; g.bias = bias; g.relu = (act == CTI_ACT_RELU);
        const int S = plan_ksplit((int)rows, out_dim, Kp, 1);
        if (S > 1) {
            g.ksplit = S;
            g.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256(planes_bytes(ra, in_dim)) + align256(planes_bytes(rb, in_dim)));
        }
        return gemm_nt_planes(g, as_stream(stream));
    }
}

extern "C" size_t cti_paralind_core_workspace_bytes(int B, int VQ, int A, int G, int K, int prec) {
    if (prec == CTI_PREC_F32 || B <= 0 || VQ <= 0 || A <= 0 || G <= 0 || K <= 0) return 0;
    return align256(planes_bytes((int64_t)B * VQ * G + PLANE_SLACK_ROWS, K)) + align256(planes_bytes((int64_t)B * A + PLANE_SLACK_ROWS, K));
}

extern "C" int cti_paralind_core_fwd(const float* M, const float* Ar, float* out, int B, int VQ, int A, int G, int K, int prec,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    CTI_REQUIRE_PTR(M); CTI_REQUIRE_PTR(Ar); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B > 0 && VQ > 0 && A > 0 && G > 0 && K > 0, CTI_E_SHAPE, "cti_paralind_core_fwd: B=%d VQ=%d A=%d G=%d K=%d", B, VQ, A, G, K);
    GemmP p{};
    p.A = M; p.B = Ar; p.C = out;
    p.lda = (int64_t)G * K; p.ldb = K; p.ldc_m = (int64_t)A * G; p.ldc_n = G;
    p.sA1 = (int64_t)VQ * G * K; p.sA2 = K;
    p.sB1 = (int64_t)A * K;      p.sB2 = 0;
    p.sC1 = (int64_t)VQ * A * G; p.sC2 = 1;
    p.nb1 = B; p.nb2 = G;
    p.M = VQ; p.N = A; p.K = K;
    p.scale = nullptr; p.scale_div = 1; p.bias = nullptr; p.relu = 0;
    if (prec == CTI_PREC_F32) return gemm_nt_f32(p, as_stream(stream));
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_paralind_core_fwd: precision mode %d", prec);
    CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(workspace_bytes >= cti_paralind_core_workspace_bytes(B, VQ, A, G, K, prec), CTI_E_WORKSPACE,
                "cti_paralind_core_fwd: workspace %zu < %zu", workspace_bytes, cti_paralind_core_workspace_bytes(B, VQ, A, G, K, prec));
    {
        const int Kp = planes_kp(K);
        const int64_t mrows = (int64_t)B * VQ * G, arows = (int64_t)B * A;
        const int64_t ra = mrows + PLANE_SLACK_ROWS, rb = arows + PLANE_SLACK_ROWS;
        unsigned short* mh = static_cast<unsigned short*>(workspace);
        unsigned short* ml = mh + (size_t)ra * Kp;
        unsigned short* ah = reinterpret_cast<unsigned short*>(static_cast<char*>(workspace) + align256(planes_bytes(ra, K)));
        unsigned short* al = ah + (size_t)rb * Kp;
        int rc = split_planes(M, K, mrows, K, mh, ml, ra, as_stream(stream)); if (rc) return rc;
        rc = split_planes(Ar, K, arows, K, ah, al, rb, as_stream(stream)); if (rc) return rc;
        PlaneGemmArgs c{};
        c.Ah = mh; c.Al = ml; c.Bh = ah; c.Bl = al; c.rows_allocA = ra; c.rows_allocB = rb;
        c.rA1 = (int64_t)VQ * G; c.rB1 = A; c.nb1 = B; c.nb2 = 1;
        c.M = VQ * G; c.N = A; c.Kp = Kp; c.terms = prec == CTI_PREC_BF16X3 ? 3 : 1; c.epi = 3; c.gdiv = G;
        c.C = out; c.ldc_m = (int64_t)A * G; c.ldc_n = G; c.sC1 = (int64_t)VQ * A * G;
        return gemm_nt_planes(c, as_stream(stream));
    }
}

// mode 3 + rank sum with M already in operand planes (cti_paralind_mbuild_planes_fwd): only Ar is split here
extern "C" size_t cti_paralind_core_planes_workspace_bytes(int B, int A, int K, int prec) {
    if (prec == CTI_PREC_F32 || B <= 0 || A <= 0 || K <= 0) return 0;
    return align256(planes_bytes((int64_t)B * A + PLANE_SLACK_ROWS, K));
}

extern "C" int cti_paralind_core_planes_fwd(const void* Mh, const void* Ml, int64_t rows_alloc, const float* Ar, float* out, int B, int VQ, int A, int G,
                                            int K, int prec, void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(Mh); CTI_REQUIRE_PTR(Ml); CTI_REQUIRE_PTR(Ar); CTI_REQUIRE_PTR(out); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(B > 0 && VQ > 0 && A > 0 && G > 0 && K > 0 && K % 32 == 0 && rows_alloc >= (int64_t)B * VQ * G + PLANE_SLACK_ROWS, CTI_E_SHAPE,
                "cti_paralind_core_planes_fwd: B=%d VQ=%d A=%d G=%d K=%d (a multiple of 32) rows_alloc=%lld", B, VQ, A, G, K, (long long)rows_alloc);
    CTI_REQUIRE(prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_paralind_core_planes_fwd: precision mode %d", prec);
    CTI_REQUIRE(workspace_bytes >= cti_paralind_core_planes_workspace_bytes(B, A, K, prec), CTI_E_WORKSPACE, "cti_paralind_core_planes_fwd: workspace too small");
    const int Kp = planes_kp(K);
    const int64_t arows = (int64_t)B * A, rb = arows + PLANE_SLACK_ROWS;
    unsigned short* ah = static_cast<unsigned short*>(workspace);
    unsigned short* al = ah + (size_t)rb * Kp;
    int rc = split_planes(Ar, K, arows, K, ah, al, rb, as_stream(stream)); if (rc) return rc;
    PlaneGemmArgs c{};
    c.Ah = static_cast<const unsigned short*>(Mh); c.Al = static_cast<const unsigned short*>(Ml); c.Bh = ah; c.Bl = al; c.rows_allocA = rows_alloc; c.rows_allocB = rb;
    c.rA1 = (int64_t)VQ * G; c.rB1 = A; c.nb1 = B; c.nb2 = 1;
    c.M = VQ * G; c.N = A; c.Kp = Kp; c.terms = prec == CTI_PREC_BF16X3 ? 3 : 1; c.epi = 3; c.gdiv = G;
    c.C = out; c.ldc_m = (int64_t)A * G; c.ldc_n = G; c.sC1 = (int64_t)VQ * A * G;
    return gemm_nt_planes(c, as_stream(stream));
}
