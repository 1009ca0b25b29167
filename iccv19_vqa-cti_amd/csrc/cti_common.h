// cti_common.h -- shared host-side helpers of libcti_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/cti_hip.h"

namespace cti {

struct F6Planes;                  // cti_f16f6.h

// thread-local last error text (the only mutable state of the library)
char* err_buf();
int fail(int code, const char* fmt, ...);

// Tuning overrides (cti_set_tuning): process-wide, meant for tests and benchmarks that must reach every tile geometry / the multi-chunk
// softmax at small shapes.  -1 / 0 = the library's own choice.  They never change results beyond fp32 summation order.
int tuning_gemm_cfg();            // -1 auto, 0 = 128x128, 1 = 256x128, 2 = 256x256 tile of the plane GEMM
int64_t tuning_tri_chunk();       // 0 auto (32768), else positions per chunk of the Tri softmax (forward and backward)
float tuning_guard_rho(int which);    // f16f6 guard policy (cti_set_tuning keys 3 / 4): cancellation estimate beyond which a call belongs in bf16x3 (0) / exact fp32 (1)
unsigned tuning_guard_poison_bits();  // key 5: status bits that NaN-fill the output
int tuning_guard_strata();            // key 7 (tests): 0 = the cancellation estimate without its strata maxima
int tuning_gru_persistent();          // key 9: 1 = cti_gru_forward may use its persistent form (cti_gru.hip)
int tuning_gemm16_sk();               // key 8: stream-K cut of cti_gemm16.hip's row products (-1 auto, 0 never, 1 wherever plannable)
int tuning_f6_core_free_cus();        // key 6: CUs the mode-3 product leaves free for the guard kernels beside it (-1 = default)

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

// ReLU as torch defines it: NaN stays NaN (fmaxf / v_max_f32 return the other operand, and a NaN in the inputs would come out of every
// layer as a clean zero -- the reference's relu propagates it, src/fc.py:24)
__device__ __forceinline__ float relu_nan(float x) { return x < 0.f ? 0.f : x; }

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

// ---- weight-norm scales of several layers in two launches (cti_paralind.hip) --------------------------------------
constexpr int WN_MAX = 48;        // layers per launch pair (the struct travels in the kernel arguments: 2.3 KiB)
constexpr int64_t WN_CHUNK = 4096;
struct WnBatch {
    int n;
    const float* wv[WN_MAX]; const float* g[WN_MAX]; float* scale[WN_MAX];
    int n_mats[WN_MAX]; int64_t elems[WN_MAX];
    int chunks_per_mat[WN_MAX]; int chunk_begin[WN_MAX + 1]; int mat_begin[WN_MAX + 1];   // filled by wn_batch_finish
};
void wn_batch_finish(WnBatch& d);
size_t wn_batch_partials(const WnBatch& d);
int wn_scale_batch(const WnBatch& d, float* partial, hipStream_t st);

// XCD-aware tile order (gfx950: 8 XCDs, private 4 MiB L2 each; workgroups are dealt round-robin, so ids congruent mod 8
// share an L2).  Each XCD walks a CONTIGUOUS chunk of the virtual tile sequence (bijective for any grid size), and inside
// one GEMM the axis with fewer tiles runs fastest, so the operand that is re-read most stays L2-resident:
// mode-3 GEMM: the 8 row tiles of M[b] (2 MiB) stay in L2 while the 25 column tiles of A^[b] stream through once;
// projection GEMMs: the 4 column tiles of the weights (L2-resident anyway) run back to back on one activation tile.
// Placement is a speed matter only: any mapping computes the same tiles.
__device__ __forceinline__ void tile_coords(int id, int total, int tiles_m, int tiles_n, int& z, int& tm, int& tn) {
    const int q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    const int tiles = tiles_m * tiles_n;
    z = vid / tiles;
    const int t = vid - z * tiles;
    if (tiles_m <= tiles_n) { tm = t % tiles_m; tn = t / tiles_m; }
    else                    { tn = t % tiles_n; tm = t / tiles_n; }
}

// ---- strided, batched NT GEMM (both operands K-contiguous) used by the projection layers and the PARALIND core
struct GemmP {
    const float* A; const float* B; float* C;
    int64_t lda, ldb, ldc_m, ldc_n;            // element strides
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;      // batch strides: z = b1 * nb2 + b2
    int nb1, nb2;
    int M, N, K;
    const float* scale; int scale_div;         // epilogue: acc * scale[b1*scale_bs + n / scale_div] (NULL = 1)
    const float* bias;                         // + bias[b1*bias_bs + n] (NULL = 0)
    int relu;
    int64_t scale_bs, bias_bs;                 // per-batch (b1) strides of scale / bias
};
int gemm_nt_f32(const GemmP& p, hipStream_t st);

// ---- the same GEMM on bf16 hi/lo planes (cti_gemm_bf16x3.hip) ----------------------------------------------------
// Planes are CHUNK-MAJOR: element (row, k) at (k >> 4) * rows_alloc * 16 + row * 16 + (k & 15); see that file's header.
struct PlaneGemmArgs {
    const unsigned short* Ah; const unsigned short* Al; const unsigned short* Bh; const unsigned short* Bl;
    // Af != NULL: the A operand is the fp32 matrix itself (rows x Kreal, row stride ldaf, 16-B aligned rows, Kreal % 4 == 0):
    // its rows are LDS-DMA'd as fp32 and split into hi/lo at fragment-read time (no split pass, no A planes); rA1/rA2 in rows.
    const float* Af; int64_t ldaf; int Kreal;
    // Abf != NULL (plain-bf16 products through cti_gemm16.hip only): the A operand is a row-major bf16 matrix (row stride ldabf elements, a multiple
    // of 8; 16-B aligned; K % 32 == 0 or zero-padded to Kp by the caller): no split pass, no A planes; rA1 / rA2 in rows
    const void* Abf; int64_t ldabf;
    int64_t rows_allocA, rows_allocB;          // allocated rows (the chunk pitch is rows_alloc * 16 elements)
    int64_t rA1, rA2, rB1, rB2;                // batch strides of the operands in ROWS
    int nb1, nb2;
    int M, N, Kp;
    int terms;                                 // 3 = bf16x3 (hi*hi + hi*lo + lo*hi), 1 = plain bf16 (hi planes only)
    int epi;                                   // 0 fp32 C, 1 hi/lo planes out, 3 fp32 C with G-interleaved rows, 4 f16f6 planes out (f6out), 5 bf16 rows out (C = bf16 storage, ldc_m / sC in elements; cti_gemm16.hip only)
    const F6Planes* f6out;                     // epi 4: output planes (logical row b1*sC1 + m -> f6_prow); Np = padded column count
    float* C; int64_t ldc_m, ldc_n, sC1, sC2;  // fp32 output (epi 0/3); for epi 1 sC1/sC2 are batch strides in plane ROWS
    unsigned short* Ph; unsigned short* Pl; int64_t rows_allocP; int Np;     // planes output (epi 1)
    int gdiv;                                  // epi 3
    const float* scale; int scale_div; const float* bias; int relu;
    int64_t scale_bs, bias_bs;                 // per-batch (b1) strides of scale / bias
    // split-K (skinny GEMMs: few output tiles, long K): ksplit > 1 runs ksplit K-ranges of Kp / ksplit as independent workgroups
    // that write fp32 partials [ksplit][M][N] into `partial`; a reduce kernel sums them and applies scale / bias / act.
    // Only for nb1 == nb2 == 1 and epi 0; pick ksplit with plan_ksplit().
    int ksplit; float* partial;
    int partials_only;                         // ksplit > 1: leave the fp32 partials [ksplit][M][N] to the caller (no reduce kernel, C / scale / bias unused)
    int64_t kc2;                               // set by the split-K path itself: batch b2 starts kc2 * b2 K-chunks into both operands
    // cti_gemm16.hip only: stream-K workspace (gemm16_sk_workspace_bytes(); zeroed once by the caller, 256-B aligned, one per stream); NULL = every tile whole
    void* sk_ws; size_t sk_ws_bytes;
};
int plan_ksplit(int M, int N, int Kp, long long nb);       // 1 = do not split
int planes_kp(int K);
size_t planes_bytes(int64_t rows_alloc, int K);
int split_planes(const float* x, int64_t ld, int64_t rows, int K, unsigned short* hi, unsigned short* lo, int64_t rows_alloc,
                 hipStream_t st);
int split_planes16(const unsigned short* x, int64_t ld, int64_t rows, int K, unsigned short* hi, unsigned short* lo, int64_t rows_alloc, hipStream_t st);   // bf16 rows -> planes (lo = 0)
int gemm_nt_planes(const PlaneGemmArgs& a, hipStream_t st);
// cti_gemm16.hip: the plain-bf16 (terms = 1) products of the 256 x 256 tile with fp32-row or planes epilogues
bool gemm16_eligible(const PlaneGemmArgs& a);
int gemm16_planes(const PlaneGemmArgs& a, hipStream_t st, int cfg = 2);      // cfg: the tile of gemm_nt_planes()'s model (0 = 128 x 128, 1 = 256 x 128, 2 = 256 x 256)
size_t gemm16_sk_workspace_bytes();
// cti_gemm_skinny.hip: few rows x long K (what callers plan a split-K for) as ONE launch: K split over a workgroup's waves, partial tiles summed in LDS
bool gemm_skinny_enabled();
bool gemm_skinny_eligible(const PlaneGemmArgs& a);
int gemm_skinny(const PlaneGemmArgs& a, hipStream_t st);
// planes of x^T for x (M x n) row-major: rows = n, depth = Mp >= M (multiple of 16, zero-filled)
int split_planes_t(const float* x, int64_t ld, int64_t M, int n, int64_t Mp, unsigned short* hi, unsigned short* lo, int64_t rows_alloc, hipStream_t st);
int plan_ksplit_tn(int64_t M, int N, int K);
constexpr int PLANE_SLACK_ROWS = 256;          // rows a GEMM tile may read past the last valid row

}  // namespace cti
