/* cti_hip.h -- C ABI of libcti_hip.so: the MI355X (gfx950) kernels of the Compact Trilinear Interaction hot path.
 *
 * The reference (aioz-ai/ICCV19_VQA-CTI) is pure Python/PyTorch and has no FFI of its own; each entry point
 * below replaces one torch op sequence of the reference's hot path (file:line given per function) and is what a
 * ctypes binding in the reference's src/{fc,tc,bc,attention,Tensor}.py would call (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller (PyTorch's caching
 *    allocator in the shipped host code), including workspaces; the library allocates nothing and keeps no
 *    process-wide mutable state: the last-error string and the test-only tuning overrides (cti_set_tuning) are thread-local;
 *  - all tensors are fp32, row-major, innermost dimension contiguous unless a stride argument says otherwise;
 *  - every launch goes to the hipStream_t passed as the last argument (`void*`, 0 = the null stream); no host
 *    synchronisation, no hipMalloc/hipFree inside (safe under hipGraph capture);
 *  - return value: 0 = success; negative = CTI_E_* argument error (nothing was launched); positive = hipError_t
 *    from the launch.  cti_last_error_string() describes the last failure on the calling thread;
 *  - the device is whatever is current on the calling thread (hipSetDevice by the caller).
 */
#ifndef CTI_HIP_H
#define CTI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTI_ABI_VERSION 1

enum {
    CTI_OK = 0,
    CTI_E_NULL = -1,      /* a required pointer is NULL            */
    CTI_E_SHAPE = -2,     /* a size is <= 0 or inconsistent        */
    CTI_E_ALIGN = -3,     /* a pointer / leading dimension is not aligned as the precision mode requires */
    CTI_E_UNSUPPORTED = -4, /* flag / mode not built                */
    CTI_E_WORKSPACE = -5  /* workspace too small                   */
};

/* arithmetic mode of the MFMA contractions */
enum {
    CTI_PREC_F32 = 0,     /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate                      */
    CTI_PREC_BF16X3 = 1,  /* fp32 operands split into bf16 hi+lo planes, 3 bf16 MFMAs per product, fp32 acc.   */
    CTI_PREC_BF16 = 2,    /* operands rounded to bf16 once, 1 MFMA per product, fp32 accumulate                 */
    CTI_PREC_F16F6 = 3    /* cti_tcnet_forward only: the mode-3 product (the dominant GEMM) and, for more than 6 answer tokens, the a side's Tucker
                             projection and rank nets as f16 hi x hi + ONE block-scaled fp6 MFMA for both cross terms (csrc/cti_f16f6.h), fp32
                             accumulate: fp32-grade at half the MFMA cycles of CTI_PREC_BF16X3, which the v / q sides keep */
};

enum { CTI_ACT_NONE = 0, CTI_ACT_RELU = 1 };

int cti_abi_version(void);
const char* cti_last_error_string(void);

/* Tuning overrides -- PER CALLING THREAD (thread-local, like the error string), for tests and benchmarks only (they select among kernels
 * that compute the same result up to fp32 summation order; the defaults are what production uses; a thread that never calls cti_set_tuning
 * always gets the defaults, whatever other threads set).  CTI_TUNE_GEMM_CFG: tile geometry of the plane GEMM (-1 = the makespan model,
 * 0 = 128x128, 1 = 256x128, 2 = 256x256).  CTI_TUNE_TRI_CHUNK: positions per chunk of the two-level Tri softmax, forward and backward
 * (0 = 32768; a multiple of 4 otherwise) -- lets a small tensor run the multi-chunk combine that BASELINE configs[1] needs
 * (1.58 M positions per sample = 49 chunks).  cti_get_tuning returns INT64_MIN for an unknown key. */
/* Guard policy of the CTI_PREC_F16F6 forward (round 5; production settings, thread-local like the rest; see "Range guard" below):
 * CTI_TUNE_GUARD_RHO_BF16X3 / CTI_TUNE_GUARD_RHO_FP32: the cancellation estimate rho, x 1000, beyond which the verdict asks for a re-run as bf16x3 /
 * as exact fp32 (defaults 2750 / 5500: the measured error laws ~1.8e-5 rho and ~0.9e-5 rho then stay a factor two under 1e-4);
 * CTI_TUNE_GUARD_POISON_BITS: which status bits NaN-fill `out` (default 31 = all; a caller that cannot re-run -- hipGraph replay -- may keep the f16f6
 * result of a call that only trips the cancellation ESTIMATE by passing 7: the status word still tells);
 * CTI_TUNE_F6_CORE_FREE_CUS: compute units the persistent mode-3 product leaves to the guard's last kernels, which run beside it on aux_stream
 * (-1 = the library's default). */
enum { CTI_TUNE_GEMM_CFG = 1, CTI_TUNE_TRI_CHUNK = 2, CTI_TUNE_GUARD_RHO_BF16X3 = 3, CTI_TUNE_GUARD_RHO_FP32 = 4, CTI_TUNE_GUARD_POISON_BITS = 5,
       CTI_TUNE_F6_CORE_FREE_CUS = 6, CTI_TUNE_GUARD_STRATA = 7 /* tests: 0 = sample evenly spaced rows only */,
       CTI_TUNE_GEMM16_SK = 8 /* cti_gemm_bf16_rows_sk: -1 = cut stream-K where it was measured to pay (three or more rounds of tiles), 0 = never, 1 = wherever a cut can be planned (tests) */,
       CTI_TUNE_GRU_PERSISTENT = 9 /* cti_gru_forward: 1 = an eligible call (CTI_PREC_BF16, save == NULL, 64 | B, 32 | H <= 1024, (H / 16) (B / 64) workgroups <= the
                                      device's compute units) runs ALL its steps as ONE launch whose workgroups meet between steps at counters in the workspace.  Such a
                                      launch needs every one of its workgroups resident: the caller must not let two of them share the device (same stream, or
                                      ordered by events); a workgroup that waits beyond CTI_GRU_PERSISTENT_SPINS polls (default 400 000) NaN-fills its outputs
                                      instead of hanging.  2 = one launch per step with the LDS-ring step kernel of rounds 3-5 (the persistent form's bit-exact reference: same
                                      accumulation order).  Default 0: one launch per step, K split over the workgroup's waves (round 6) */ };
int cti_set_tuning(int key, int64_t value);
int64_t cti_get_tuning(int key);

/* hipEvent_t helpers for hosts without HIP bindings (timing on the launch stream). elapsed_ms synchronises on `end`. */
void* cti_event_create(void);
int cti_event_destroy(void* ev);
int cti_event_record(void* ev, void* stream);
int cti_event_elapsed_ms(void* begin, void* end, float* ms);

/* ---- FCNet: weight_norm(Linear, dim=None) [+ReLU]  (reference src/fc.py:20-29,33-34) ------------------------- */

/* scale[i] = g[i] / ||V_i||_F for n_mats matrices stored back to back, `elems` floats each.
 * Replaces the weight_norm pre-forward hook (torch `_weight_norm`: norm, div, mul over the whole weight) that
 * src/fc.py:22,27 installs; the scaled weight itself is never materialised (the scale is a GEMM-epilogue factor). */
int cti_wn_scale(const float* weight_v, const float* weight_g, float* scale, int n_mats, int64_t elems, void* workspace,
                 size_t workspace_bytes, void* stream);
size_t cti_wn_scale_workspace_bytes(int n_mats, int64_t elems);
/* The same for n layers of different sizes (n_mats = 1 each) in two launches per 48 layers: scale[i][0] = weight_g[i][0] / ||weight_v[i]||_F over
 * elems[i] floats.  The four arrays are HOST arrays of device pointers / sizes (consumed before the call returns); workspace of
 * cti_wn_scale_many_workspace_bytes(elems, n). */
int cti_wn_scale_many(const float* const* weight_v, const float* const* weight_g, float* const* scale, const int64_t* elems, int n, void* workspace,
                      size_t workspace_bytes, void* stream);
size_t cti_wn_scale_many_workspace_bytes(const int64_t* elems, int n);   /* workspace may be NULL (one workgroup per matrix: slow for >16K elements) */

/* y[r, n] = act( scale[n / scale_div] * sum_k x[r, k] * w[n, k] + bias[n] )          (src/fc.py:33-34, nn.Linear)
 * x: rows x in_dim, row stride ldx;  w: out_dim x in_dim (weight_v), row stride ldw;  y: rows x out_dim, row stride ldy.
 * scale: device array of ceil(out_dim / scale_div) floats (scale_div = out_dim for one FCNet layer; = h/rank for the
 * rank nets of src/tc.py:29-31 packed as one out_dim = h matrix); NULL = 1.  bias: NULL = 0.
 * prec: CTI_PREC_*.  For CTI_PREC_BF16X3 / CTI_PREC_BF16 `workspace` must hold cti_wn_linear_workspace_bytes(). */
int cti_wn_linear_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* scale, int scale_div,
                      const float* bias, float* y, int64_t ldy, int64_t rows, int in_dim, int out_dim, int act,
                      int prec, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_wn_linear_workspace_bytes(int64_t rows, int in_dim, int out_dim, int prec);

/* ---- zero-row mask  (src/attention.py:36,55: `0 == v.abs().sum(2)`) ------------------------------------------- */
/* mask[r] = 1 iff every element of row r is +-0 (bit-exact statement of the reference's test for finite input). */
int cti_zero_row_mask(const float* v, int64_t ldv, uint8_t* mask, int64_t rows, int dim, void* stream);
/* The same mask of a bf16 matrix (round 5: BASELINE configs[2] / [3] name bf16 tensors): every element +-0.  dim, ldv even. */
int cti_zero_row_mask_bf16(const void* v_bf16, int64_t ldv, uint8_t* mask, int64_t rows, int dim, void* stream);

/* ---- PARALIND core  (src/tc.py:46-52 -> src/Tensor.py:3-20) ---------------------------------------------------- */

/* T_eff[r,i,j,k,g] = T[r,i,<scrambled (j,k,g)>]: the fixed index scramble src/Tensor.py:6-8 applies for G > 1
 * (SURVEY.md 3.4).  src: (R,I,J,K,G) contiguous (= T_g (1,R,hr,hr,hr,G,1) with I=J=K=hr); dst: same shape.
 * inverse != 0 scatters instead of gathers (maps a T_eff-shaped gradient back to T_g's layout). */
int cti_teff_scramble(const float* src, float* dst, int R, int I, int J, int K, int G, int inverse, void* stream);

/* M[b,v,q,g,r*K+k] = sum_ij T_eff[r,i,j,k,g] * Vr[b,v,r*I+i] * Qr[b,q,r*J+j]
 * (mode-1 and mode-2 products of src/Tensor.py:6-13; r stays in the contraction index of the mode-3 GEMM).
 * Vr: (B,V,R*I), Qr: (B,Q,R*J) contiguous (outputs of the packed rank nets); M: (B,V,Q,G,R*K) contiguous. */
int cti_paralind_mbuild_fwd(const float* Vr, const float* Qr, const float* Teff, float* M, int B, int V, int Q,
                            int R, int I, int J, int K, int G, void* stream);
/* The same M, written directly as the bf16 hi/lo operand planes of the mode-3 GEMM (chunk-major [R*hr/16][rows_alloc][16], see
 * DESIGN.md section 3; R*hr must be a multiple of 32, rows_alloc >= B*V*Q*G + 256).  Teff_t: NULL, or T_eff with the two inner
 * axes swapped to [r][(j,k,g)][i] (cti_transpose_f32 per r) -- with it, hr = 16, G = 2, V <= 48, Q <= 16 run on the MFMA. */
int cti_paralind_mbuild_planes_fwd(const float* Vr, const float* Qr, const float* Teff, const float* Teff_t, unsigned short* Mh,
                                   unsigned short* Ml, int B, int V, int Q, int R, int hr, int G, int64_t rows_alloc, void* stream);

/* The same M, written directly as the f16f6 operand block of the mode-3 product in the f16f6 mode (below: cti_f16f6_planes_bytes(B*V*Q*G, R*hr,
 * V*Q*G); rows (b,v,q,g) in batches of V*Q*G).  Teff_t as above (required).  hr = 16, G = 2, even R and (2560 + 128 Q) V <= 160 KiB of LDS
 * (BASELINE configs[1]: V = 36, Q = 14); CTI_E_UNSUPPORTED otherwise (cti_paralind_mbuild_fwd + cti_quantize_f16f6 give the same block).
 * Slack / padding rows are left untouched; values beyond f16's range (outside the format's domain) clamp to +-65504. */
int cti_paralind_mbuild_f16f6_fwd(const float* Vr, const float* Qr, const float* Teff_t, void* planes, size_t planes_bytes, int B, int V, int Q,
                                  int R, int hr, int G, void* stream);

/* out[b,vq,a,g] = sum_K M[b,vq,g,K] * Ar[b,a,K]   (mode-3 product + the sum over ranks, src/Tensor.py:16-20 and the
 * running `+ f_emb` of src/tc.py:50).  M: (B,VQ,G,K); Ar: (B,A,K); out: (B,VQ,A,G) contiguous = the logical
 * (B,V,Q,A,G) tensor TCNet.forward returns.  prec / workspace as for cti_wn_linear_fwd. */
int cti_paralind_core_fwd(const float* M, const float* Ar, float* out, int B, int VQ, int A, int G, int K,
                          int prec, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_paralind_core_workspace_bytes(int B, int VQ, int A, int G, int K, int prec);

/* ---- TCNet.forward as one call  (src/tc.py:41-52) --------------------------------------------------------------- */
/* out[b,v,q,a,g] = TCNet.forward(v, q, a) in eval mode (dropout = identity): 3 Tucker projections, the 3 x R rank nets
 * (passed PACKED: rank_wv[s] is (h, h) = the R weight_v matrices (h/R, h) stacked, rank_g[s] (R), rank_b[s] (h)),
 * T_eff scramble, modes 1+2, mode 3 + rank sum.  Arrays of 3 are in (v, q, a) order and live on the HOST; the pointers
 * in them are device pointers.  tucker_wv[s]: (h, in_s) contiguous, tucker_g[s]: scalar, tucker_b[s]: (h).
 * v (B,V,v_dim), q (B,Q,q_dim), a (B,A,a_dim), out (B,V,Q,A,G) contiguous.  zero_mask: NULL or (B,V) bytes, filled as
 * cti_zero_row_mask(v) (what TriAttention needs next).  All intermediates live in `workspace`
 * (cti_tcnet_forward_workspace_bytes); between MFMA GEMMs they stay bf16 hi/lo planes (no fp32 round trip).
 * ev_core_begin / ev_core_end: NULL, or hipEvent_t handles (cti_event_create) recorded on `stream` immediately before and
 * after the mode-3 GEMM launch -- how bench.py measures the dominant kernel inside the timed region.
 * aux_stream: NULL, or a second hipStream_t of the caller: the v/q-side chain + M build then run on it beside the a-side chain
 * (fork/join by events inside the call; on return all work is ordered behind `stream` as usual).
 * prepared: NULL (weight-norm scales, T_eff and the weights' operand planes are recomputed in this call), or the block written by
 * cti_tcnet_prepare for the SAME weights, widths and precision: inference then holds its weights in GEMM-operand form and the call
 * starts with the activations. */
int cti_tcnet_forward(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                      const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                      const float* const* rank_g, const float* const* rank_b, const float* T_g, float* out,
                      uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                      int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                      void* ev_core_end, void* aux_stream, void* stream);
size_t cti_tcnet_forward_workspace_bytes(int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                                         int G, int prec);
/* cti_tcnet_forward that also leaves the PARTIAL PASS of TriAttention's softmax (src/attention.py:55-58) behind: every wave of the mode-3
 * GEMM reduces its own accumulators to (max, sum exp(x - max)) per glimpse over the outputs whose v row is not all-zero (zero_mask,
 * required here), so that the softmax reads the (B,V,Q,A,G) logits ONCE (cti_masked_softmax_tri_from_partials_fwd) instead of twice.
 * `out` itself is written unmasked, exactly as by cti_tcnet_forward.  sm_partials: a device block of cti_tcnet_softmax_partials_bytes(...)
 * bytes, laid out [B][chunk][G][2]; that function returns 0 where the fused pass does not exist (today: outside prec = CTI_PREC_F16F6,
 * G = 2, h % 32 == 0) -- callers then use cti_tcnet_forward + cti_masked_softmax_tri_fwd. */
int cti_tcnet_forward_sm(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                         const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                         const float* const* rank_g, const float* const* rank_b, const float* T_g, float* out,
                         uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                         int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                         void* ev_core_end, void* aux_stream, void* stream, float* sm_partials, size_t sm_partials_bytes);
size_t cti_tcnet_softmax_partials_bytes(int B, int V, int Q, int A, int h, int G, int prec);

/* Range guard of the CTI_PREC_F16F6 forward.  The reference multiplies in full-range fp32 (src/Tensor.py:12,18); the f16f6 operand format is
 * fp32-grade only for magnitudes inside f16's normal range (csrc/cti_f16f6.h: 6.1e-5 <= |x| <= 65504).  Whenever cti_tcnet_forward /
 * cti_tcnet_forward_sm run their f16f6 kernels -- cti_tcnet_forward_guard_bytes(...) != 0 -- the first that many bytes of `workspace` are
 * the guard block: a scan of every encoded operand (`a`, its Tucker projection, A^, M, the a-side weights: the per-block scale bytes the
 * encoders wrote) and of V^ / Q^ / T_eff for non-finite values leaves a status word -- without an aux_stream on `stream` BEFORE ev_core_begin is
 * recorded; with one (round 5) its last kernels run on aux_stream BESIDE the mode-3 product (which leaves them a few compute units) and are the LAST
 * work the call enqueues there: an event the caller records on aux_stream after the call returns marks the verdict --
 * (uint32 at offset 0):  0 = every operand is inside the format's domain;  CTI_GUARD_SATURATED = a block reaches beyond +-61440 (incl.
 * inf / NaN in an encoded tensor);  CTI_GUARD_UNDERFLOW = a tensor that is not all zero has no block above 2^-12 (its f16 hi parts are
 * subnormal);  CTI_GUARD_NONFINITE = inf / NaN in V^, Q^ or T_eff.  With a non-zero status the call still completes, but `out` is then
 * OVERWRITTEN WITH NaN (stream-ordered, no host involvement: valid under hipGraph capture) -- a caller never receives clamped numbers.
 * cti_guard_read waits on the HOST for the event it is given -- ev_core_begin of a call without aux_stream, else an event recorded on aux_stream
 * after the call (either way the mode-3 product is still running) --, copies the status word through `stream` (a stream of the caller's that is
 * idle by then) and returns it: a non-zero value means "re-run this call with CTI_PREC_BF16X3" (what the shipped Python wrapper does).  It is the
 * only entry point of the library that synchronises. */
/* Accuracy (round 4): inside the range the f16f6 product's error is ~2^-17 sum_k |M_k A^_k| per output (2^-15 worst case) against fp32's 2^-24: it
 * is 1e-4 of the LARGEST output only while the contraction does not cancel too heavily.  The guard therefore also samples 32 x 32 (M row, A^ row)
 * pairs per batch from the f16 planes -- per operand 16 evenly spaced rows and the LARGEST row (by its blocks' scale bytes) of each of 16 strata, so
 * that a few outsized answer tokens cannot hide (round 5) -- and forms  rho = max sum_k |m_k a_k| / max |sum_k m_k a_k|  (the measured error of the
 * whole forward is at most ~1.8e-5 rho as f16f6, ~0.9e-5 rho as bf16x3, ~3e-7 rho in fp32; 1.4-1.8 on the synthetic BASELINE tensors);
 * CTI_GUARD_CANCEL: rho > 2.75 -- re-run with CTI_PREC_BF16X3;  CTI_GUARD_CANCEL_HEAVY: rho > 5.5 -- re-run with CTI_PREC_F32 (CTI_TUNE_GUARD_RHO_*).
 * Numerator and denominator are maxima over ALL batches of the call, as the tolerance is (1e-4 of the tensor's largest output).
 * cti_guard_read_ratio copies rho (diagnostics / tests; it synchronises `stream`). */
enum { CTI_GUARD_SATURATED = 1, CTI_GUARD_UNDERFLOW = 2, CTI_GUARD_NONFINITE = 4, CTI_GUARD_CANCEL = 8, CTI_GUARD_CANCEL_HEAVY = 16 };
int cti_guard_read_ratio(const void* workspace, void* stream, float* ratio_host);
size_t cti_tcnet_forward_guard_bytes(int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R, int G, int prec);
int cti_guard_read(const void* workspace, void* event, void* stream, uint32_t* status_host);
/* The batch-independent part of cti_tcnet_forward, computed once per parameter update: the six weight-norm scales, T_eff (and its
 * [r][(j,k,g)][i] copy), and in the bf16 modes the hi/lo operand planes of the six weight matrices.  `prepared`: a device block of
 * cti_tcnet_prepared_bytes(...) owned by the caller; valid until a weight, T_g or the precision changes. */
int cti_tcnet_prepare(const float* const* tucker_wv, const float* const* tucker_g, const float* const* rank_wv, const float* const* rank_g,
                      const float* T_g, int v_dim, int q_dim, int a_dim, int h, int R, int G, int prec, void* prepared, size_t prepared_bytes,
                      void* stream);
size_t cti_tcnet_prepared_bytes(int v_dim, int q_dim, int a_dim, int h, int R, int G, int prec);

/* TriAttention.forward (reference src/attention.py:49-59) as ONE call: logits = TCNet.forward(v, q, a); rows v with every element == 0 get
 * -inf (zero_mask, required, receives the mask); p[b,:,:,:,g] = softmax over the flattened (v, q, a) axis.  Arguments as cti_tcnet_forward plus
 * p_out (B,V,Q,A,G).  Few answer tokens (A <= 6 with h/R = 16, glimpse 2: the FFOE / MC models): the fused modes-1+2+3 kernel holds a sample's
 * logits in registers and writes `logits` and `p_out` itself -- no softmax launch at all; CTI_PREC_F16F6 with glimpse 2: the mode-3 product
 * leaves the softmax's partial pass and ONE normalise pass follows; otherwise the two-pass masked softmax runs behind the product.
 * workspace: cti_triattention_workspace_bytes(...) (its head is cti_tcnet_forward's workspace, range-guard block included).
 * v_tucker_out (nullable; only where cti_triattention_hoist_ok(...) == 1): relu(v_tucker(v)) computed by the caller -- hoisted into the batched
 * GEMM of the glimpses' pooling networks (one read of v for t_att and every t_net[g]) -- as fp32 rows (B / v_rep * V, h) with row
 * stride ld_vt floats; v_rep > 1: batch rows b * v_rep .. b * v_rep + v_rep - 1 carry the SAME image (the MC pipeline repeats every image per
 * candidate answer, src/MC/train.py:75-79) and the v side runs once per image.  `v` itself is still read for the zero-row mask. */
size_t cti_triattention_workspace_bytes(int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R, int G, int prec);
int cti_triattention_forward(const float* v, const float* q, const float* a, const float* const* tucker_wv,
                             const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                             const float* const* rank_g, const float* const* rank_b, const float* T_g, float* logits, float* p_out,
                             uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                             int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                             void* ev_core_end, void* aux_stream, void* stream, const float* v_tucker_out, int64_t ld_vt, int v_rep);
int cti_triattention_hoist_ok(int B, int V, int Q, int A, int h, int R, int G, int prec);
/* Round 5 -- bf16 activations (BASELINE configs[2] / [3] name bf16 tensors): the same call with `v` (B, V, v_dim) AND the hoisted projection v_tucker_out as
 * bf16 rows (ld_vt in elements, a multiple of 8; what cti_gemm_bf16_rows writes with c_bf16 = 1).  v is read for the zero-row mask only; requires
 * cti_triattention_hoist_ok(...) and a hoisted projection -- CTI_E_UNSUPPORTED otherwise (widen v, call cti_triattention_forward). */
int cti_triattention_forward_vt16(const void* v_bf16, const float* q, const float* a, const float* const* tucker_wv,
                                  const float* const* tucker_g, const float* const* tucker_b, const float* const* rank_wv,
                                  const float* const* rank_g, const float* const* rank_b, const float* T_g, float* logits, float* p_out,
                                  uint8_t* zero_mask, int B, int V, int Q, int A, int v_dim, int q_dim, int a_dim, int h, int R,
                                  int G, int act, int prec, const void* prepared, void* workspace, size_t workspace_bytes, void* ev_core_begin,
                                  void* ev_core_end, void* aux_stream, void* stream, const void* v_tucker_out_bf16, int64_t ld_vt, int v_rep);

/* ---- masked softmax  (src/attention.py:55-58 Tri, :35-39 Bi) --------------------------------------------------- */

/* Tri: logits (B, V, QA, G) contiguous, G innermost.  In place: rows v with mask[b,v] != 0 are filled with -inf
 * (the reference's `logits.data.masked_fill_`), then p[b,:,g] = softmax over the flattened (v,qa) axis per (b,g).
 * An all-masked sample yields NaN like the reference.  workspace: cti_softmax_tri_workspace_bytes(). */
int cti_masked_softmax_tri_fwd(float* logits, const uint8_t* mask, float* p, int B, int V, int64_t QA, int G,
                               void* workspace, size_t workspace_bytes, void* stream);
/* The same result from the partials cti_tcnet_forward_sm left (partials_bytes = what cti_tcnet_softmax_partials_bytes returned): combine,
 * then ONE pass over the logits that fills -inf on masked rows and writes p.  G = 2; logits / p 16-B aligned.  workspace: B*G*2 floats. */
int cti_masked_softmax_tri_from_partials_fwd(float* logits, const uint8_t* mask, const float* partials, size_t partials_bytes, float* p,
                                             int B, int V, int64_t QA, int G, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_softmax_tri_workspace_bytes(int B, int V, int64_t QA, int G);

/* Bi: logits (B, G, V, Q) contiguous; mask (B,V) or NULL (v_mask=False); p[b,g,:] = softmax over (v,q). */
int cti_masked_softmax_bi_fwd(float* logits, const uint8_t* mask, float* p, int B, int G, int V, int Q, void* stream);

/* ---- attention-weighted sum-pools  (src/tc.py:59 einsum, src/bc.py:73 + :75-77) -------------------------------- */

/* out[b,d] = sum_{v,q,a} vt[b,v,d] * w[b,v,q,a] * qt[b,q,d] * at[b,a,d].  vt/qt/at contiguous (B,*,D);
 * w is addressed with element strides (w_sb, w_sv, w_sq, w_sa) so that the caller can pass `att[..., g]`. */
int cti_tri_pool_fwd(const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                     int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, void* stream);
/* The same pool on the MFMA (fp32-grade 3-product bf16 mode) where that form is the faster one: A = 6 (the MC model's answer length),
 * Q <= 16, V <= 64, D % 32 == 0; returns CTI_E_UNSUPPORTED without a message otherwise (call cti_tri_pool_fwd then).
 * v_rep >= 1: vt holds ONE (V, D) block per image and batch rows b * v_rep .. b * v_rep + v_rep - 1 share it (the MC pipeline repeats
 * every image per candidate answer, src/MC/train.py:75-79) -- vt is then (B / v_rep, V, D). */
int cti_tri_pool_mfma_fwd(const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                          int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, int v_rep, void* stream);

/* The "shifted" pools of the hoisted glimpse loops (inference; base_model.py `_HoistedLoop`, replacing the per-glimpse q_net / q_tucker / a_tucker
 * GEMMs of src/FFOE/base_model.py:53-61,129-132 and src/MC/base_model.py:145-148): the q (and a) operand is relu(row + add[b, :]) formed as the
 * rows are loaded -- row = the pre-activation projection of the INITIAL sequence (one batched GEMM for all glimpses), add = the projection of the
 * residual accumulated so far, one (B, D) vector per sample, or NULL = 0.
 *   tri: out[b,d] = sum_vqa vt[b / v_rep, v, d] w[b,v,q,a] relu(qt[b,q,d] + qadd[b,d]) relu(at[b,a,d] + aadd[b,d]);  use_mfma: 1 = the fp32-grade MFMA form
 *        where it applies, 2 = the same with ONE bf16 product per pair (plain-bf16 mode), 0 = the VALU kernels (v_rep must then be 1);   bi (k = 1): out[b,d] = sum_vq vt[b,v,d] w[b,v,q] relu(qt[b,q,d] + qadd[b,d]).
 * CTI_E_UNSUPPORTED (nothing launched, no message) when no kernel with the on-load shift takes the shape: materialise the operands instead. */
int cti_tri_pool_shift_fwd(const float* vt, const float* qt, const float* at, const float* qadd, const float* aadd, const float* w,
                           int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, int v_rep,
                           int use_mfma, void* stream);
int cti_bi_pool_shift_fwd(const float* vt, const float* qt, const float* qadd, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                          float* out, int B, int V, int Q, int D, void* stream);
/* Round 5: the shifted pools with `vt` -- the tensor these kernels exist to stream: (B, V, D) per glimpse -- as bf16 ROWS, what the plain-bf16 mode's hoisted
 * v projection writes (cti_gemm_bf16_rows, c_bf16 = 1): half the bytes.  qt / at / w / out stay fp32.  The tri form exists on the MFMA kernel only (use_mfma != 0,
 * A = 3 or 6); CTI_E_UNSUPPORTED (nothing launched, no message) otherwise. */
int cti_tri_pool_shift_vt16_fwd(const void* vt_bf16, const float* qt, const float* at, const float* qadd, const float* aadd, const float* w,
                                int64_t w_sb, int64_t w_sv, int64_t w_sq, int64_t w_sa, float* out, int B, int V, int Q, int A, int D, int v_rep,
                                int use_mfma, void* stream);
/* Round 5 -- the unrolled BAN glimpse loop (base_model.py `_ban_forward_unrolled`; reference src/FFOE/base_model.py:53-61): the pool's shift as a SUM of n_qadd
 * addends, qadds[i] a (B, qadd_ld[i]) fp32 matrix read at columns [0, D) (row stride 0 = one row for the whole batch), summed as the pool loads them:
 *   out[b,d] = sum_vq vt[b,v,d] w[b,v,q] relu(qt[b,q,d] + sum_i qadds[i][b * qadd_ld[i] + d]).
 * The addends are the raw split-K slabs of the products that feed the shift (cti_gemm_pb_partials): no reduce launch between a product and the pool.
 * qadds / qadd_ld: HOST arrays, consumed before the call returns; addends 16-B aligned, ld % 4 == 0, at most 32.  vt_bf16 != 0: vt as bf16 rows.
 * out: row stride ldo >= D (the pooled vectors of all glimpses side by side = the K-concatenated operand of the loop's last product).
 * CTI_E_UNSUPPORTED (nothing launched, no message) outside the streaming kernel's shapes (Q <= 16, D % 4 == 0). */
int cti_bi_pool_shift_multi_fwd(const void* vt, int vt_bf16, const float* qt, const float* const* qadds, const int64_t* qadd_ld, int n_qadd, const float* w,
                                int64_t w_sb, int64_t w_sv, int64_t w_sq, float* out, int64_t ldo, int B, int V, int Q, int D, void* stream);
/* partials[s][m][n] = the s-th K range's share of x (M, K; row stride ldx) @ W^T, W (N, K) as cti_split_operand planes: the RAW fp32 slabs of the split-K plan,
 * s < cti_gemm_pb_partials_count(M, N, K) (>= 1) -- no reduce pass, no scale, no bias.  workspace: cti_gemm_pb_partials_workspace_bytes(M, K, prec). */
int cti_gemm_pb_partials_count(int M, int N, int K);
size_t cti_gemm_pb_partials_workspace_bytes(int M, int K, int prec);
int cti_gemm_pb_partials(const float* x, int64_t ldx, const void* W_planes, int M, int N, int K, int prec, float* partials, size_t partials_bytes, void* workspace,
                         size_t workspace_bytes, void* stream);
int cti_bi_pool_shift_vt16_fwd(const void* vt_bf16, const float* qt, const float* qadd, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                               float* out, int B, int V, int Q, int D, void* stream);

/* out[b,n] = sum_{t<k} sum_{v,q} vt[b,v,n*k+t] * w[b,v,q] * qt[b,q,n*k+t],  n < D/k.  w == NULL means w = 1
 * (that is BCNet.forward with h_out=None, src/bc.py:42-47: out is then (B,1,D) with k = 1). */
int cti_bi_pool_fwd(const float* vt, const float* qt, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                    float* out, int B, int V, int Q, int D, int k, void* stream);

/* ---- bilinear attention logits  (src/bc.py:52-58; also :63-68 with h = weight-normalised h_net) ---------------- */
/* logits[b,g,v,q] = h_scale[0] * sum_d vt[b,v,d] * h[g,d] * qt[b,q,d] + h_bias[g].
 * h: (G,D) (= h_mat_v or h_net.weight_v); h_scale: device scalar (g/||h||_F from cti_wn_scale) or NULL = 1;
 * h_bias: (G) or NULL.  The (B,G,V,D) intermediate of src/bc.py:55 is never materialised. */
int cti_bi_logits_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                      float* logits, int B, int G, int V, int Q, int D, void* stream);
/* The same logits on the MFMA in the fp32-grade 3-product bf16 mode (h[g,d] * qt[b,q,d] formed on the fly, fragments loaded straight
 * from global memory).  Returns CTI_E_UNSUPPORTED without a message when D % 16 != 0 or an operand is not 16-B aligned: call
 * cti_bi_logits_fwd then. */
int cti_bi_logits_mfma_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                           float* logits, int B, int G, int V, int Q, int D, void* stream);
/* cti_bi_logits_mfma_fwd in the arithmetic of `prec`: CTI_PREC_BF16 = ONE bf16 product per pair (the plain-bf16 mode of the model forwards: half the operand
 * splitting, a third of the MFMAs), any other value the fp32-grade three-product form.  Same shapes, same CTI_E_UNSUPPORTED rule. */
int cti_bi_logits_prec_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                           float* logits, int B, int G, int V, int Q, int D, int prec, void* stream);
/* ... with `vt` as bf16 rows (round 5): the LDS-staged kernel's shapes only (V <= 64, G*Q <= 128, D % 32 == 0); CTI_E_UNSUPPORTED otherwise. */
int cti_bi_logits_prec_vt16_fwd(const void* vt_bf16, const float* qt, const float* h, const float* h_scale, const float* h_bias,
                                float* logits, int B, int G, int V, int Q, int D, int prec, void* stream);

/* BiAttention.forward_all's logits + mask + softmax in ONE launch (round 3; reference src/attention.py:29-40 on the projections of src/bc.py:52-57): the
 * bilinear logits as cti_bi_logits_mfma_fwd, then the last workgroup to add into a sample's logits fills the rows of `mask` ((B, V) bytes, 1 = all-zero object
 * row; NULL = no mask) with -inf in `logits` and writes p = softmax over (V, Q) per glimpse (an all-masked sample gives the reference's NaN row).
 * `counters`: B ints, ZERO at entry and zero again at exit -- allocate and zero them once.  Returns CTI_E_UNSUPPORTED (nothing launched, no message) outside
 * V <= 64, Q <= 16, G*Q <= 128, D % 32 == 0 and 16-B aligned operands: the caller then takes cti_bi_logits_(mfma_)fwd + cti_masked_softmax_bi_fwd. */
int cti_biattention_fwd(const float* vt, const float* qt, const float* h, const float* h_scale, const float* h_bias, const uint8_t* mask,
                        float* logits, float* p, int* counters, int B, int G, int V, int Q, int D, void* stream);

/* ---- backward-pass primitives ----------------------------------------------------------------------------------------
 * The gradient contractions of every layer are NT GEMMs of transposed operands; these entry points are what the
 * autograd Functions of the host code (iccv19_vqa-cti_amd/autograd.py) are made of.  The reference has no explicit
 * backward code: torch.autograd differentiates src/fc.py, src/tc.py, src/bc.py, src/attention.py op by op. */

/* C[z][m,n] = act(scale[b1*scale_bs + n/scale_div] * sum_k A[z][m,k] * B[z][n,k] + bias[b1*bias_bs + n]).  A is ONE row-major matrix of
 * rowsA_total x K (row stride lda); batch z = (b1, b2), b1 < nb1, b2 < nb2 uses its rows [b1*rA1 + b2*rA2, +M); likewise B
 * with N rows.  C element (m,n) of batch z at C[b1*sC1 + b2*sC2 + m*ldc_m + n*ldc_n]. */
int cti_gemm_nt(const float* A, int64_t lda, int64_t rowsA_total, int64_t rA1, int64_t rA2, const float* B, int64_t ldb,
                int64_t rowsB_total, int64_t rB1, int64_t rB2, float* C, int64_t ldc_m, int64_t ldc_n, int64_t sC1,
                int64_t sC2, int nb1, int nb2, int M, int N, int K, const float* scale, int scale_div, int64_t scale_bs,
                const float* bias, int64_t bias_bs, int act, int prec, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_gemm_nt_workspace_bytes(int64_t rowsA_total, int64_t rowsB_total, int K, int prec);

/* Resident operand planes: split a (rows x K) fp32 matrix -- a weight -- into the GEMM's bf16 hi/lo operand layout ONCE (block of
 * cti_operand_planes_bytes), then run any number of products against it with cti_gemm_nt_pb (= cti_gemm_nt, nb2 = 1, contiguous C rows,
 * whose B operand is that block; batch b1 uses rows [b1*rB1, +N) of it).  bf16 modes only. */
size_t cti_operand_planes_bytes(int64_t rows, int K);
int cti_split_operand(const float* x, int64_t ld, int64_t rows, int K, void* planes, size_t planes_bytes, void* stream);
int cti_gemm_nt_pb(const float* A, int64_t lda, int64_t rowsA_total, int64_t rA1, const void* B_planes, int64_t rowsB_total, int64_t rB1,
                   float* C, int64_t ldc_m, int64_t sC1, int nb1, int M, int N, int K, const float* scale, int scale_div, int64_t scale_bs,
                   const float* bias, int64_t bias_bs, int act, int prec, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_gemm_nt_pb_workspace_bytes(int64_t rowsA_total, int64_t rowsB_total, int K, int prec);    /* an upper bound for any (nb1, M, N) on these rows */
/* ... and what ONE call needs exactly: the A planes + the fp32 partials of the split-K plan cti_gemm_nt_pb makes for (nb1, M, N) (round 5) */
size_t cti_gemm_nt_pb_workspace_bytes2(int64_t rowsA_total, int64_t rowsB_total, int K, int prec, int nb1, int M, int N);
/* The same product in the plain-bf16 arithmetic with A given as a row-major bf16 matrix (row stride lda elements, a multiple of 8; 16-B aligned; K % 32
 * == 0): read as it stands -- no split pass, no workspace -- by the round-4 kernel (csrc/cti_gemm16.hip: 256 x 256 / 288 x 192 tiles, two wave groups one
 * interval apart).  C: fp32 rows (c_bf16 = 0; 16-B aligned, ldc_m % 4 == 0) or bf16 rows (c_bf16 = 1; 8-B aligned): the output of one layer of reference
 * src/fc.py:22-29 as the next layer's A operand.  BASELINE configs[2] / [3] ("bf16") run their projections through this. */
int cti_gemm_bf16_rows(const void* A_bf16, int64_t lda, int64_t rowsA_total, int64_t rA1, const void* B_planes, int64_t rowsB_total, int64_t rB1,
                       void* C, int c_bf16, int64_t ldc_m, int64_t sC1, int nb1, int M, int N, int K, const float* scale, int scale_div,
                       int64_t scale_bs, const float* bias, int64_t bias_bs, int act, void* stream);
/* Round 6: the same product cut STREAM-K (csrc/cti_gemm16.hip, "Stream-K") when its T output tiles do not fill whole rounds of the P compute units
 * (T > P, T % P != 0, K >= 512: e.g. the hoisted projections of reference src/fc.py:22-29 at 9 216 x 3 072 = 432 tiles = 1.69 rounds): the last one-to-two
 * rounds' tiles become ONE sequence of K stages shared out evenly; a tile then has at most two contributors, the first of which leaves its accumulators in
 * `workspace` for the second to start from -- bit-identical to the uncut product (same additions in the same order).  `workspace`:
 * cti_gemm_bf16_rows_sk_workspace_bytes() bytes, 256-B aligned, ZEROED ONCE by the caller (the kernel leaves its flag words at zero), never shared by two
 * calls that may run at the same time (one per stream); word [compute units] of it is an error word the kernel sets if a contributor's flag never came
 * (bounded poll; cannot happen while workgroups are dispatched in order).  workspace = NULL: every tile whole (= cti_gemm_bf16_rows). */
size_t cti_gemm_bf16_rows_sk_workspace_bytes(void);
int cti_gemm_bf16_rows_sk(const void* A_bf16, int64_t lda, int64_t rowsA_total, int64_t rA1, const void* B_planes, int64_t rowsB_total, int64_t rB1,
                          void* C, int c_bf16, int64_t ldc_m, int64_t sC1, int nb1, int M, int N, int K, const float* scale, int scale_div,
                          int64_t scale_bs, const float* bias, int64_t bias_bs, int act, void* workspace, size_t workspace_bytes, void* stream);

/* The "f16f6" operand format (csrc/cti_f16f6.h): an fp32 matrix as an f16 hi plane plus ONE block-scaled fp6 (e2m3) plane -- the codes of the
 * residual -- and two E8M0 scales per 32 elements (of the hi part, whose fp6 codes the GEMM derives in registers, and of the residual): 2.81
 * bytes per element -- so that a product costs one f16 MFMA and half a (4x-rate) fp6 MFMA:
 * a*b ~= a16*b16 + fp6(a16)*fp6(b - b16) + fp6(a - a16)*fp6(b16), fp32 accumulate (fp32-grade: ~2e-5 normalised on the mode-3 product).
 * cti_quantize_f16f6 encodes `rows` x K fp32 (row stride ld) into a caller-owned block of cti_f16f6_planes_bytes (256-B aligned);
 * batch_rows > 0 places every batch of batch_rows rows at a multiple of 8 plane rows (what batched products need), 0 = one matrix.
 * cti_gemm_nt_f16f6: C[z][m,n] = act(scale[n/scale_div] * sum_k A[z][m,k] B[z][n,k] + bias[n]) on two such blocks; gdiv > 1 interleaves the
 * GEMM rows (row m' = m*gdiv + g -> C[z*sC + m*ldc_m + g + n*ldc_n]), the mode-3 product of src/Tensor.py:16-20 with G = gdiv. */
size_t cti_f16f6_planes_bytes(int64_t rows, int K, int64_t batch_rows);
int cti_quantize_f16f6(const float* x, int64_t ld, int64_t rows, int K, int64_t batch_rows, void* planes, size_t planes_bytes, void* stream);
/* The same encoding pass WITHOUT the zero fill of the block's slack / padding rows that cti_quantize_f16f6 issues first: for a block that call has filled
 * before (or the caller zeroed) -- the pass exactly as cti_tcnet_forward runs it on `a` (src/tc.py:43's input), which is what a benchmark of it should time. */
int cti_quantize_f16f6_into(const float* x, int64_t ld, int64_t rows, int K, int64_t batch_rows, void* planes, size_t planes_bytes, void* stream);
int cti_gemm_nt_f16f6(const void* A_planes, int64_t rowsA_total, int64_t batch_rowsA, const void* B_planes, int64_t rowsB_total, int64_t batch_rowsB,
                      float* C, int64_t ldc_m, int64_t ldc_n, int64_t sC, int gdiv, int nb, int M, int N, int K, const float* scale, int scale_div,
                      const float* bias, int act, void* stream);
/* A weight-normalised Linear layer between two f16f6 operands (src/fc.py:22-29 as the a-side rank nets of src/tc.py:46 run it in the f16f6
 * mode): Y = act(X W^T + bias[m]) for X (N rows x K) and W (M features x K) given as f16f6 blocks, Y (N x M) written as an f16f6 block
 * (cti_f16f6_planes_bytes(N, M, batch_rows_out); batches of batch_rows_out rows start at multiples of 8 plane rows).  The product is taken
 * as W X^T so that a lane pair of the accumulator holds one row's 32-feature block and the encoder runs in registers; the bias is the
 * accumulators' initial value.  The weight-norm scale g / ||V|| belongs in W's block: cti_quantize_f16f6_scaled multiplies row m by
 * row_scale[m / scale_div] while encoding.  M % 32 == 0; slack / padding rows of Y are left untouched. */
int cti_gemm_nt_f16f6_planes(const void* W_planes, int64_t rowsW_total, const void* X_planes, int64_t rowsX_total, void* Y_planes, size_t Y_bytes,
                             int64_t batch_rows_out, int M, int N, int K, const float* bias, int act, void* stream);
int cti_quantize_f16f6_scaled(const float* x, int64_t ld, int64_t rows, int K, int64_t batch_rows, const float* row_scale, int scale_div, void* planes,
                              size_t planes_bytes, void* stream);

/* C (N x K, contiguous) = a^T b for a (M x N, row stride lda) and b (M x K, row stride ldb): the weight-gradient contraction over the
 * ROW axis (dW = dz^T x; src/fc.py:22-29 under autograd).  Both operands are written straight to transposed bf16 hi/lo planes, the M axis
 * is split over extra workgroups and a reduce kernel sums the partials.  prec = BF16X3 or BF16 (the exact-fp32 mode uses
 * cti_transpose_f32 + cti_gemm_nt). */
int cti_gemm_tn(const float* a, int64_t lda, const float* b, int64_t ldb, float* C, int64_t M, int N, int K, int prec, void* workspace,
                size_t workspace_bytes, void* stream);
size_t cti_gemm_tn_workspace_bytes(int64_t M, int N, int K, int prec);
/* C (rows x K) = a (rows x N) . b (N x K), row-major with leading dimensions lda / ldb, C contiguous: the input gradient dx = dzs . V of a
 * Linear layer (torch.autograd of src/fc.py:22-29) without a transposed copy of the weight.  bf16 modes only. */
int cti_gemm_nn(const float* a, int64_t lda, const float* b, int64_t ldb, float* C, int64_t rows, int N, int K, int prec, void* workspace,
                size_t workspace_bytes, void* stream);
size_t cti_gemm_nn_workspace_bytes(int64_t rows, int N, int K, int prec);

/* dst[b][c][r] = src[b][r][c] (fp32, `batch` matrices of rows x cols). */
int cti_transpose_f32(const float* src, int64_t ld_src, int64_t batch_stride_src, float* dst, int64_t ld_dst,
                      int64_t batch_stride_dst, int rows, int cols, int batch, void* stream);

/* dst[i] = alpha * sum_b src[b*n + i] + beta * dst[i]   (split-K partial sums, per-chunk column sums). */
int cti_sum_batches(const float* src, float* dst, int nb, int64_t n, float alpha, float beta, void* stream);

/* dst[c] = alpha * sum_r src[r, c] + beta * dst[c] for a tall contiguous (rows, n) matrix (bias gradients: the GRU's b_ih / b_hh),
 * two stages over row groups; workspace of cti_col_sum_workspace_bytes(rows, n). */
int cti_col_sum(const float* src, int64_t rows, int n, float* dst, float alpha, float beta, void* workspace, size_t workspace_bytes,
                void* stream);
size_t cti_col_sum_workspace_bytes(int64_t rows, int n);

/* Backward of y = act(scale * u + bias) w.r.t. u (src/fc.py:24,29 ReLU):  dzs[r,n] = scale[n/scale_div] * dy[r,n] * (y[r,n] > 0)
 * (act = RELU; without activation the (y > 0) factor is dropped);  dbias[n] = sum_r dy[r,n] * (y[r,n] > 0) (unscaled).
 * y, dy, dzs: rows x n contiguous. */
int cti_act_bwd(const float* dy, const float* y, const float* scale, int scale_div, float* dzs, float* dbias, int64_t rows, int n,
                int act, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_act_bwd_workspace_bytes(int64_t rows, int n);

/* nn.Dropout (src/fc.py:20-21,25-26; src/bc.py:29).  use_mask = 0: draw keep ~ Bernoulli(1-p) from Philox-4x32-10 keyed by
 * `seed`, counter = offset + element/4; store it in mask[i] (1 byte) and write y = x * keep / (1-p).  use_mask = 1: reuse the
 * stored mask (the backward pass: x = dy).  x == y is allowed.  period > 0: x is read at i % period, i.e. y is n / period
 * independently masked copies of x (the R rank nets of src/tc.py:29-31 each draw their own mask of the shared input).
 * y == NULL with use_mask = 0: only the mask is drawn (x is not read) -- the input of cti_ranknets_drop_*. */
int cti_dropout(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, int use_mask,
                int64_t period, void* stream);

/* Backward of cti_paralind_core_fwd (mode 3 + rank sum, src/Tensor.py:16-20, src/tc.py:50) as two streaming passes, exact fp32:
 * dM[b,v,q,g,k] = sum_a dout[b,v,q,a,g] Ar[b,a,k];  dAr[b,a,k] = sum_{v,q,g} dout[b,v,q,a,g] M[b,v,q,g,k].  dout (B,V,Q,A,G), M and dM
 * (B,V,Q,G,K), Ar and dAr (B,A,K), all contiguous.  K % 4 != 0, A > 8 or operands off 16-B alignment: CTI_E_UNSUPPORTED, no message. */
int cti_paralind_core_bwd(const float* dout, const float* M, const float* Ar, float* dM, float* dAr, int B, int V, int Q, int A, int G, int K,
                          void* stream);

/* The same pair with M held as the bf16 hi/lo operand planes cti_paralind_mbuild_planes_fwd writes (element (row, k) at
 * [k >> 4][row][k & 15], rows_alloc rows per 16-column chunk) -- the TRAINING forward keeps M only in this form: the mode-3 GEMM reads it
 * without a split pass, the backward rebuilds M = hi + lo for dAr.  K % 32 == 0; fwd: prec = CTI_PREC_BF16X3 or CTI_PREC_BF16,
 * workspace of cti_paralind_core_planes_workspace_bytes (the planes of Ar); bwd: A <= 8. */
int cti_paralind_core_planes_fwd(const void* Mh, const void* Ml, int64_t rows_alloc, const float* Ar, float* out, int B, int VQ, int A, int G, int K,
                                 int prec, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_paralind_core_planes_workspace_bytes(int B, int A, int K, int prec);
int cti_paralind_core_bwd_planes(const float* dout, const void* Mh, const void* Ml, int64_t rows_alloc, const float* Ar, float* dM, float* dAr, int B,
                                 int V, int Q, int A, int G, int K, void* stream);

/* The R rank nets of TCNet in train mode (src/tc.py:29-31, 44-46 with src/fc.py:25-28: every FCNet([h, hr]) drops its OWN mask on the
 * shared input) without materialising the R masked copies: mask (R, rows, h) bytes from cti_dropout(y = NULL), x (rows, h), W (R*hr, h) the
 * packed weight_v, scale (R,) from cti_wn_scale, bias (R*hr,) or NULL, p the drop probability.
 *   fwd: y[m, r*hr+n]  = act(scale[r]/(1-p) * sum_k x[m,k] mask[r,m,k] W[r*hr+n,k] + bias[r*hr+n])            y (rows, R*hr)
 *   dw : G[r*hr+n, k]  = 1/(1-p) * sum_m dzs[m, r*hr+n] x[m,k] mask[r,m,k]                (input of cti_wn_bwd)  G (R*hr, h)
 *   dx : dx[m, k]      = 1/(1-p) * sum_r mask[r,m,k] sum_n dzs[m, r*hr+n] W[r*hr+n, k]                          dx (rows, h)
 * dzs (rows, R*hr) is cti_act_bwd's output.  fp32 MFMA (exact products) in every precision mode.  Shapes outside hr <= 16, h % 4 == 0
 * (fwd: h <= 512) or unaligned operands return CTI_E_UNSUPPORTED without a message: the caller takes cti_dropout(period) + cti_gemm_nt. */
int cti_ranknets_drop_fwd(const float* x, const uint8_t* mask, const float* W, const float* scale, const float* bias, float* y, int64_t rows,
                          int h, int R, int hr, float p, int relu, void* stream);

/* cti_ranknets_drop_fwd on the bf16 matrix cores (round 3): the products as bf16 hi / lo split products on the 16x16x16 MFMA (prec = CTI_PREC_BF16X3: three per
 * pair, fp32-grade; CTI_PREC_BF16: one), x split once per wave and masked per rank in registers, W pre-split into `workspace`
 * (cti_ranknets_drop_fwd_mfma_workspace_bytes(h, R, hr) bytes, 16-B aligned) and shared by a workgroup's row tiles through LDS.  Same arguments and result as
 * cti_ranknets_drop_fwd.  Returns CTI_E_UNSUPPORTED (nothing launched, no message) unless h = 512 and hr = 16 (the reference's --h_mm 512 --rank 32,
 * src/FFOE/main.py:61-64), for prec = CTI_PREC_F32 and for unaligned operands: the caller then takes cti_ranknets_drop_fwd. */
size_t cti_ranknets_drop_fwd_mfma_workspace_bytes(int h, int R, int hr);
int cti_ranknets_drop_fwd_mfma(const float* x, const uint8_t* mask, const float* W, const float* scale, const float* bias, float* y,
                               int64_t rows, int h, int R, int hr, float p, int relu, int prec, void* workspace, size_t workspace_bytes, void* stream);
int cti_ranknets_drop_dw(const float* dzs, const float* x, const uint8_t* mask, float* G, int64_t rows, int h, int R, int hr, float p,
                         void* stream);
int cti_ranknets_drop_dx(const float* dzs, const float* W, const uint8_t* mask, float* dx, int64_t rows, int h, int R, int hr, float p,
                         void* stream);

/* Gradient through weight_norm(dim=None) (torch `_weight_norm` backward): W = g * V / ||V||_F = s * V.  Given
 * G = dzs^T x (n_mats matrices of `elems` floats, the gradient w.r.t. V through the direct path):
 * dweight_g[i] = <G_i, V_i> / g_i;  dweight_v = G - (<G,V> / ||V||^2) * V. */
int cti_wn_bwd(const float* G, const float* weight_v, const float* weight_g, float* dweight_v, float* dweight_g, int n_mats,
               int64_t elems, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_wn_bwd_workspace_bytes(int n_mats, int64_t elems);   /* 0 for matrices of <= 16,384 elements (one workgroup each) */

/* Backward of cti_paralind_mbuild_fwd (hr = I = J = K in {4, 8, 16}).  dM (B,V,Q,G,R*hr); dVr (B,V,R*hr); dQr (B,Q,R*hr);
 * dTeff_partial (B,R,hr,hr,hr,G): per-sample partials of dT_eff -- sum them with cti_sum_batches, then map to T_g's layout with
 * cti_teff_scramble(inverse = 1). */
int cti_paralind_mbuild_bwd(const float* dM, const float* Vr, const float* Qr, const float* Teff, float* dVr, float* dQr,
                            float* dTeff_partial, int B, int V, int Q, int R, int hr, int G, void* stream);

/* Softmax backward, dlogits = p * (dp - sum_axis p * dp).  Tri: p, dp, dlogits (B, V*QA, G), axis = V*QA per (b,g).
 * Bi: rows x N contiguous, axis = N.  Masked positions have p = 0 and therefore receive 0. */
int cti_masked_softmax_tri_bwd(const float* p, const float* dp, float* dlogits, int B, int V, int64_t QA, int G, void* workspace,
                               size_t workspace_bytes, void* stream);
size_t cti_softmax_tri_bwd_workspace_bytes(int B, int V, int64_t QA, int G);
int cti_masked_softmax_bi_bwd(const float* p, const float* dp, float* dlogits, int rows, int N, void* stream);

/* Backward of cti_tri_pool_fwd: dvt (B,V,D), dqt (B,Q,D), dat (B,A,D), dw (B,V,Q,A) contiguous or NULL. */
int cti_tri_pool_bwd(const float* dout, const float* vt, const float* qt, const float* at, const float* w, int64_t w_sb, int64_t w_sv,
                     int64_t w_sq, int64_t w_sa, float* dvt, float* dqt, float* dat, float* dw, int B, int V, int Q, int A, int D,
                     void* stream);
/* The attention gradient of either pool on the MFMA (fp32-grade 3-product bf16 mode), k = 1:
 *   dw[b,v,q,a] = sum_d dout[b,d] vt[b,v,d] qt[b,q,d] at[b,a,d]      (at == NULL, A = 1: the bi pool, dw (B,V,Q)).
 * Returns CTI_E_UNSUPPORTED without a message when D % 16 != 0 or an operand is not 16-B aligned: pass dw to cti_*_pool_bwd then. */
int cti_pool_dw_mfma(const float* dout, const float* vt, const float* qt, const float* at, float* dw, int B, int V, int Q, int A, int D,
                     void* stream);
/* Backward of cti_bi_pool_fwd: dout (B, D/k); dvt (B,V,D), dqt (B,Q,D), dw (B,V,Q) contiguous or NULL. */
int cti_bi_pool_bwd(const float* dout, const float* vt, const float* qt, const float* w, int64_t w_sb, int64_t w_sv, int64_t w_sq,
                    float* dvt, float* dqt, float* dw, int B, int V, int Q, int D, int k, void* stream);
/* Backward of cti_bi_logits_fwd.  dvt (B,V,D), dqt (B,Q,D); dh_partial (B,G,D) = per-sample partials of h_scale * dL/d(h_scale*h)
 * (sum over B = the `G` argument of cti_wn_bwd, or dL/dh itself when h_scale is NULL); dh_bias_partial (B,G). */
int cti_bi_logits_bwd(const float* dlogits, const float* vt, const float* qt, const float* h, const float* h_scale, float* dvt,
                      float* dqt, float* dh_partial, float* dh_bias_partial, int B, int G, int V, int Q, int D, void* stream);

/* out[r] = sum_c x[r, c] for a contiguous (rows, cols) matrix, one wave per row (the bias gradient of the bilinear logits). */
int cti_row_sum(const float* x, float* out, int64_t rows, int cols, void* stream);

/* dvt, dqt and dh_partial of cti_bi_logits_bwd on the MFMA (fp32-grade mode) for G <= 8, V <= 64, Q <= 16, D % 32 == 0; the bias partial
 * stays with cti_bi_logits_bwd's row sums.  Returns CTI_E_UNSUPPORTED without a message otherwise. */
int cti_bi_logits_bwd_mfma(const float* dlogits, const float* vt, const float* qt, const float* h, const float* h_scale, float* dvt,
                           float* dqt, float* dh_partial, int B, int G, int V, int Q, int D, void* stream);

/* ---- data-parallel update -------------------------------------------------------------------------------
 * What follows the single RCCL all-reduce of the flat gradient buffer; replaces Trainer._all_reduce_and_rescale + _opt
 * (src/FFOE/trainer.py:221-269), utils.clip_grad_norm_ (src/utils.py:323-328) and torch.optim.Adamax.step (src/FFOE/train.py:34).
 * grad *= inv_denom in place, partial[0..1023] = per-workgroup sums of squares (workspace of cti_optim_workspace_bytes()). */
int cti_flat_scale_sumsq(float* grad, int64_t n, float inv_denom, float* partial, void* stream);
/* Packs per-parameter gradient tensors into the flat buffer in one launch (replaces Trainer._get_flat_grads' torch.cat,
 * src/FFOE/trainer.py:245-255).  table: HOST array of n_entries x 3 int64 {source address (0 = no gradient), first float of the
 * parameter's slot in flat, element count}, sorted by slot start, slots disjoint and starting on multiples of 4 floats;
 * flat[slot + j] = src[j]; every float outside the sources (alignment padding, parameters without a gradient) is set to 0; an entry
 * whose source already IS its slot is left untouched.  n: floats in flat, a multiple of 4.  The entries travel in kernel arguments
 * (160 per launch), so the host array may be reused as soon as the call returns. */
int cti_flat_gather(const int64_t* table, int n_entries, float* flat, int64_t n, void* stream);
/* norm = sqrt(sum partial); coef = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no clipping); g' = coef * grad;
 * exp_avg = b1*exp_avg + (1-b1)*g'; exp_inf = max(b2*exp_inf, |g'| + eps); param -= lr / (1 - b1^step) * exp_avg / exp_inf.
 * grad_norm_out: NULL or one float (the pre-clip norm, for logging without a blocking .item() in the step). */
int cti_adamax_step(float* param, const float* grad, float* exp_avg, float* exp_inf, int64_t n, const float* partial, float max_norm,
                    float lr, float beta1, float beta2, float eps, int step, float* grad_norm_out, void* stream);
size_t cti_optim_workspace_bytes(void);

/* ---- rows either side of the CTI path (N1 model forward, N3 GRU + word embedding, N4 classifier + losses) ----
 * WordEmbedding.forward (src/language_model.py:40-46): out[i, 0:dim] = table0[tokens[i]], and out[i, dim:2*dim] =
 * table1[tokens[i]] when table1 != NULL ('c' in op: the trainable and the frozen table, concatenated).  tokens: int64;
 * tables: (rows, dim) contiguous; a token outside [0, rows) produces NaNs (torch raises IndexError there). */
int cti_embedding_fwd(const int64_t* tokens, const float* table0, const float* table1, float* out, int64_t n, int dim, int64_t rows,
                      void* stream);
/* The same lookup as bf16 rows of pitch ld_out (>= the row's width, zero-filled beyond it; reference src/language_model.py:40-46): the operand
 * cti_gru_forward_x16 reads as it stands (round 6: the plain-bf16 mode's word vectors never exist as fp32 rows). */
int cti_embedding_fwd_bf16(const int64_t* tokens, const float* table0, const float* table1, void* out_bf16, int64_t ld_out, int64_t n, int dim, int64_t rows,
                           void* stream);
/* dtable[tokens[i], :] += dout[i, col_off : col_off + dim] (atomic adds; the caller zeroes dtable); row padding_idx receives
 * nothing, like nn.Embedding(padding_idx = ntoken) (src/language_model.py:19). */
int cti_embedding_bwd(const int64_t* tokens, const float* dout, int64_t ld_dout, int col_off, float* dtable, int64_t n, int dim,
                      int64_t rows, int64_t padding_idx, void* stream);

/* nn.GRU(in, H, 1, batch_first=True) from a zero state (src/language_model.py:57-61,91-96; gate order r, z, n), every step in one
 * call: x (B,T,I), w_ih (3H,I), w_hh (3H,H), b_ih / b_hh (3H) -> out (B,T,H) = all hidden states.  save: NULL, or (T,B,5,H) =
 * (r, z, n, W_hn h + b_hn, h_t) per step, what cti_gru_backward needs.  The input projection is one GEMM over all steps; a step is
 * one split-K GEMM against pre-split recurrent weights + one gate kernel that also emits h_t as the next GEMM's bf16 planes.
 * w_ih_planes / w_hh_planes: NULL, or resident planes of the two weight matrices (cti_split_operand; bf16 modes). */
int cti_gru_forward(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* out, float* save,
                    int B, int T, int I, int H, int prec, const void* w_ih_planes, const void* w_hh_planes, void* workspace, size_t workspace_bytes,
                    void* stream);
/* cti_gru_forward with x as bf16 rows: (B * T) rows of pitch ldx = I rounded up to a multiple of 32, zero beyond column I, 16-B aligned base
 * (what cti_embedding_fwd_bf16 writes).  CTI_PREC_BF16 only: the input-side product reads the rows as they stand -- no split pass. */
int cti_gru_forward_x16(const void* x_bf16, int64_t ldx, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* out, float* save,
                        int B, int T, int I, int H, int prec, const void* w_ih_planes, const void* w_hh_planes, void* workspace, size_t workspace_bytes,
                        void* stream);
size_t cti_gru_forward_workspace_bytes(int B, int T, int I, int H, int prec);
/* Back-propagation through time.  dout (B,T,H) contiguous; writes the pre-activation gradients dgi (B,T,3H) (input side: dx = dgi W_ih,
 * dW_ih = dgi^T x, db_ih = column sums) and dgh (T,B,3H) (hidden side, time-major: dW_hh = sum_t dgh_t^T h_{t-1}, db_hh = column
 * sums); the caller forms those products with cti_gemm_nt / cti_col_sum. */
int cti_gru_backward(const float* dout, const float* w_hh, const float* save, float* dgi, float* dgh, int B, int T, int H, int prec,
                     void* workspace, size_t workspace_bytes, void* stream);
size_t cti_gru_backward_workspace_bytes(int B, int T, int H, int prec);

/* cti_paralind_mbuild_bwd on the matrix cores (round 3; reference: the autograd of src/Tensor.py:9-14 through src/tc.py:48-50): the five contractions of
 * a rank as 16x16x16 bf16 MFMA products with operands split into bf16 hi + lo in registers (prec = CTI_PREC_BF16X3: three products per pair, fp32-grade;
 * CTI_PREC_BF16: one).  Same arguments and outputs as cti_paralind_mbuild_bwd, except that dTeff_partial holds cti_paralind_mbuild_bwd_mfma_partials(B, R) partials
 * instead of B (a workgroup owns a rank and a chunk of samples and sums its chunk's dT_eff in registers).  Returns CTI_E_UNSUPPORTED -- nothing launched, no message --
 * outside hr = 16, G = 2, V <= 48, Q <= 16, V*Q*G <= 1024 or its LDS budget, for prec = CTI_PREC_F32, or for dM / Teff that are not 16-B aligned: the caller
 * then takes cti_paralind_mbuild_bwd (exact fp32). */
int cti_paralind_mbuild_bwd_mfma_partials(int B, int R);   /* n: dTeff_partial is (n, R, hr, hr, hr, G) here -- one partial per chunk of samples, summed in registers */
int cti_paralind_mbuild_bwd_mfma(const float* dM, const float* Vr, const float* Qr, const float* Teff, float* dVr, float* dQr,
                                 float* dTeff_partial, int B, int V, int Q, int R, int hr, int G, int prec, void* stream);

/* M-build backward for any cubic core size h/rank (plain fp32 VALU kernels through the forward's intermediates; cti_paralind_mbuild_bwd is the
 * fast form for h/rank in {4, 8, 16} and returns CTI_E_UNSUPPORTED otherwise).  Same outputs: dVr, dQr, per-sample dT_eff partials (B, R,hr,hr,hr,G). */
int cti_paralind_mbuild_bwd_generic(const float* dM, const float* Vr, const float* Qr, const float* Teff, float* dVr, float* dQr, float* dTeff_partial,
                                    int B, int V, int Q, int R, int hr, int G, void* workspace, size_t workspace_bytes, void* stream);
size_t cti_paralind_mbuild_bwd_generic_workspace_bytes(int B, int V, int R, int hr, int G);

/* hipGraph-safe forms (a captured training step replays with the same kernel arguments): whatever changes from step to step lives in DEVICE
 * memory.  cti_dropout_g = cti_dropout whose Philox key is advanced by rng_dev[0] (NULL = cti_dropout); cti_adamax_step_g = cti_adamax_step with
 * the learning rate and the number of COMPLETED steps read from device memory (bias correction 1 - beta1^(steps_done + 1));
 * cti_counter_add bumps such a counter in stream order (after the kernels that read it). */
int cti_dropout_g(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, int use_mask, int64_t period,
                  const uint64_t* rng_dev, void* stream);
int cti_adamax_step_g(float* param, const float* grad, float* exp_avg, float* exp_inf, int64_t n, const float* partial, float max_norm,
                      const float* lr_dev, float beta1, float beta2, float eps, const int64_t* steps_done_dev, float* grad_norm_out, void* stream);
int cti_counter_add(int64_t* counter, int64_t inc, void* stream);

/* Swish (src/activation.py:17-22), the classifier's alternative activation (src/classifier.py:14). */
int cti_swish_fwd(const float* x, float* y, int64_t n, void* stream);
int cti_swish_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);

/* out[b,h] = beta * out[b,h] + sum_l x[b,l,h]          (q_emb.sum(1), src/FFOE/base_model.py:66,134; x (B,L,H) contiguous) */
int cti_seq_sum(const float* x, float* out, int B, int L, int H, float beta, void* stream);
/* out[b,l,h] = x[b,l,h] + y[b,h]  (x NULL = 0)         (q_prj(b_emb.unsqueeze(1)) + q_emb, src/FFOE/base_model.py:61,131-132) */
int cti_seq_bcast_add(const float* x, const float* y, float* out, int B, int L, int H, void* stream);
/* Replicated batch rows (the MC pipeline feeds every image once per candidate answer, src/MC/train.py:75-79; `TanModel.v_replication = 'auto'`):
 * cti_rows_equal_prev: eq[b] = 1 when rows b and b - 1 of x (row_bytes each, a multiple of 16, 16-B aligned; else CTI_E_UNSUPPORTED) hold the same bits, eq[0] = 0.
 * cti_poison_unless_replicated: out[0..n) = NaN unless eq[b] == 1 for every b with b % r != 0 -- a forward that ran under the assumption "groups of r identical
 * rows" gives NaNs on a batch that breaks it, never a plausible wrong answer. */
int cti_rows_equal_prev(const void* x, int64_t row_bytes, int B, unsigned char* eq, void* stream);
int cti_poison_unless_replicated(const unsigned char* eq, int B, int r, float* out, int64_t n, void* stream);
/* y[r, n] = act(scale[0] * sum_k x[r, k] W[n, k] + bias[n]) in exact fp32 for N <= 8 outputs (the MC models' answer head: src/classifier.py:26 with out_dim = 2,
 * `weight_norm(Linear)` as scale = g / ||V||_F); scale / bias may be NULL; relu != 0 applies max(., 0).  CTI_E_UNSUPPORTED for N > 8 (use cti_wn_linear_fwd). */
int cti_linear_small_n(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* scale, const float* bias, float* y, int64_t ldy, int rows, int K, int N,
                       int relu, void* stream);
/* out[b,h] = cq * sum_l q[b,l,h] + ca * sum_l a[b,l,h] + dq * Dq[b,h] + da * Da[b,h]; q (B,Lq,H), a (B,La,H) or NULL, Dq / Da (B,H) or NULL: the classifier
 * input of the hoisted glimpse loops in one pass (src/FFOE/base_model.py:63-64,134). */
int cti_joint_sums(const float* q, int Lq, float cq, const float* a, int La, float ca, const float* Dq, float dq, const float* Da, float da, float* out, int B, int H,
                   void* stream);
/* out[i] = a * x[i] + b * y[i], i < n (out may alias x or y). */
int cti_axpby(const float* x, float a, const float* y, float b, float* out, int64_t n, void* stream);
/* The residual projection of a glimpse as ONE call (src/FFOE/base_model.py:61,131-132 `q_prj(b_emb.unsqueeze(1)) + q_emb`, and the sequence sums
 * of :66,134): y = scale * (x @ W^T) + bias with x (B, K) fp32 and W given as resident planes (cti_split_operand of the (N x K) weight_v);
 * out[b,l,:] = seq[b,l,:] + y[b,:]; acc (nullable) [b,:] = beta * acc[b,:] + sum_l out[b,l,:].  Three launches (split of x, split-K GEMM,
 * one fused reduce + broadcast-add + sum pass) instead of five.  N % 4 == 0; bf16 modes only (the planes); workspace:
 * cti_linear_residual_workspace_bytes(B, N, K, prec). */
size_t cti_linear_residual_workspace_bytes(int B, int N, int K, int prec);
int cti_linear_residual_pb(const float* x, int64_t ldx, const void* W_planes, const float* scale, int scale_div, const float* bias, const float* seq,
                           float* out, float* acc, float beta, int B, int L, int N, int K, int prec, void* workspace, size_t workspace_bytes, void* stream);

/* nn.BCEWithLogitsLoss(reduction='sum') per row (src/FFOE/train.py:28-33, divided by the batch size at src/FFOE/trainer.py:189-190):
 * row_loss[r] = sum_c max(x,0) - x*t + log(1 + exp(-|x|)); reduce the rows with cti_sum_batches.
 * Backward: dx = beta * dx + coef * (*upstream) * (sigmoid(x) - t)   (upstream: device scalar or NULL = 1). */
int cti_bce_logits_rows_fwd(const float* x, const float* target, float* row_loss, int rows, int n, void* stream);
int cti_bce_logits_bwd(const float* x, const float* target, const float* upstream, float coef, float* dx, int64_t n, float beta,
                       void* stream);
/* Distillation term (src/loss_function.py:21-24): row_kl[r] = sum_c pk (log pk - log ps), pk = softmax(knowledge/T), ps = softmax(x/T).
 * Backward: dx = beta * dx + coef * (*upstream) / T * (ps - pk). */
int cti_kd_rows_fwd(const float* x, const float* knowledge, float* row_kl, int rows, int n, float T, void* stream);
int cti_kd_rows_bwd(const float* x, const float* knowledge, const float* upstream, float coef, float* dx, int rows, int n, float T,
                    float beta, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTI_HIP_H */
