// cti_api.hip -- error reporting and the C-ABI entry points that dispatch to the MFMA GEMMs.
#include "cti_common.h"
#include <string.h>

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

}  // namespace cti

using namespace cti;

extern "C" int cti_abi_version(void) { return CTI_ABI_VERSION; }
extern "C" const char* cti_last_error_string(void) { return err_buf(); }

extern "C" size_t cti_wn_linear_workspace_bytes(int64_t rows, int in_dim, int out_dim, int prec) {
    (void)rows; (void)in_dim; (void)out_dim;
    if (prec == CTI_PREC_F32) return 0;
    return 0;
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
    switch (prec) {
        case CTI_PREC_F32: return gemm_nt_f32(p, as_stream(stream));
        default: return fail(CTI_E_UNSUPPORTED, "cti_wn_linear_fwd: precision mode %d is not built", prec);
    }
}

extern "C" size_t cti_paralind_core_workspace_bytes(int B, int VQ, int A, int G, int K, int prec) {
    (void)B; (void)VQ; (void)A; (void)G; (void)K;
    if (prec == CTI_PREC_F32) return 0;
    return 0;
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
    switch (prec) {
        case CTI_PREC_F32: return gemm_nt_f32(p, as_stream(stream));
        default: return fail(CTI_E_UNSUPPORTED, "cti_paralind_core_fwd: precision mode %d is not built", prec);
    }
}
