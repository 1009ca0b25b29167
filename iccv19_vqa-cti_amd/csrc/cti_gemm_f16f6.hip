// cti_gemm_f16f6.hip -- fp32-grade NT GEMM on f16 + block-scaled fp6 matrix-core products (format and rationale: cti_f16f6.h).
//
// Per 32-wide K block and 32x32 output tile: two v_mfma_f32_32x32x16_f16 (hi x hi) and ONE v_mfma_scale_f32_32x32x64_f8f6f4
// whose lane halves carry the two cross terms (lanes 0-31: fp6(A_hi) x fp6(B_lo), lanes 32-63: fp6(A_lo) x fp6(B_hi)), all into
// the same fp32 accumulator: 96 MFMA cycles per block and tile against 192 for the three-bf16-product form.
//
// KERNEL.  256 x 192 tile, 8 waves (4 x 2, each 64 x 96 = 2 x 3 MFMA tiles, 96 accumulator registers), a 3-slot LDS ring of
// 32-deep K blocks (52 KiB per slot: f16 rows of 64 B, fp6 rows of 24 B for the hi and the lo codes, scale bytes), filled by
// LDS-DMA with counted vmcnt and one raw s_barrier per block (two blocks in flight behind the MFMAs), persistent XCD-aware tile
// walk -- the ring protocol of cti_gemm_bf16x3.hip.  LDS images: the f16 rows are XOR-swizzled at the SOURCE (16-B chunk c of row
// r lands at chunk c ^ ((r >> 2) & 3)): conflict-free ds_read_b128 fragments; the fp6 rows are linear (24-B pitch: three
// conflict-free ds_read_b64 per lane).
#include "cti_common.h"
#include "cti_f16f6.h"

namespace cti {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));   // 8-B LDS loads.  NOT HIP's uint2 / uint8_t: loads through struct or char types make hipcc
                                                              // drain vmcnt(0) -- every LDS-DMA in flight -- in front of them (may-alias with the DMA's LDS store)

// ---- encoder: fp32 rows -> f16f6 planes.  A wave stages a 64-row x 32-column tile through a private LDS patch: the global loads are
// coalesced (8 lanes x 16 B = one row's block, 8 rows per instruction), then every lane encodes one row from LDS (pitch 36 floats:
// conflict-free 16-B row reads) with the streaming encoder the GEMM epilogues use.
__global__ __launch_bounds__(256) void quantize_f16f6_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int K, F6Planes p) {
    __shared__ __attribute__((aligned(16))) float patch[4][64 * 36];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int kb = blockIdx.y, k0 = kb * 32;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * 64;
    if (row0 >= rows) return;
    float* st = patch[wid];
    const int c4 = lane & 7, rsub = lane >> 3;
    const bool vec = (k0 + 32 <= K) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rr = it * 8 + rsub;
        const int64_t row = row0 + rr;
        f6_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < rows) {
            const float* src = x + row * ld + k0 + c4 * 4;
            if (vec) v = *reinterpret_cast<const f6_f32x4*>(src);
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (k0 + c4 * 4 + u < K) v[u] = src[u];
            }
        }
        *reinterpret_cast<f6_f32x4*>(st + rr * 36 + c4 * 4) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // same wave writes and reads its patch: in-order LDS, no barrier
    const int64_t row = row0 + lane;
    if (row < rows) f6_encode_row32_lds(st + lane * 36, p, f6_prow(p, row), kb);
}

// ---- GEMM --------------------------------------------------------------------------------------------------------------
struct F6P {
    const char* AH; const char* AFH; const char* AFL; const char* AS;
    const char* BH; const char* BFH; const char* BFL; const char* BS;
    int64_t pA, pAS, pB, pBS;                  // rows_alloc / rows_allocS of the two operands
    int64_t rA, rB;
    int M, N, Kb, total_tiles;
    float* C; int64_t ldc_m, ldc_n, sC; int gdiv;
    const float* scale; int scale_div; const float* bias; int relu;
    F6Planes P; int Np;
    int desync_ticks;                          // 100 MHz ticks over which the workgroups' first tiles are spread (see the kernel)
};
enum { F6_EPI_F32 = 0, F6_EPI_INTERLEAVE2 = 2, F6_EPI_INTERLEAVE = 3, F6_EPI_PLANES = 4 };

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WM_, int WN_, int TM_, int TN_, int NST_>
struct GeoF {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, NST = NST_;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NTHR = WM * WN * 64, NW = WM * WN;
    // pieces (1 KiB LDS-DMA wave-instructions) per slot: H rows of 64 B, FH / FL rows of 24 B (rounded up), one piece of scales
    static constexpr int PAH = BM / 16, PAF = (BM * 24 + 1023) / 1024, PBH = BN / 16, PBF = (BN * 24 + 1023) / 1024;
    static constexpr int OFF_AH = 0, OFF_AFH = OFF_AH + PAH * 1024, OFF_AFL = OFF_AFH + PAF * 1024, OFF_AS = OFF_AFL + PAF * 1024;
    static constexpr int OFF_BH = OFF_AS + 1024, OFF_BFH = OFF_BH + PBH * 1024, OFF_BFL = OFF_BFH + PBF * 1024, OFF_BS = OFF_BFL + PBF * 1024;
    static constexpr int SLOT = OFF_BS + 1024;
    static constexpr int NP = SLOT / 1024;                          // pieces per slot
    static constexpr int LDS = NST * SLOT;
    static_assert(LDS <= 160 * 1024, "ring exceeds the CU's LDS");
};

#ifndef CTI_F6_NT_B
#define CTI_F6_NT_B 0            // 1: the B operand's pieces (streamed once per XCD) are loaded non-temporally so that they do not displace the A tiles in L2
#endif
template <int AUX = 0>
__device__ __forceinline__ void dma16(const char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

#ifndef CTI_F6_ABL          // timing-only ablations (wrong results): 1 no refill DMA, 2 no MFMA, 4 no fragment reads, 8 no epilogue stores
#define CTI_F6_ABL 0
#endif

// One LDS-DMA piece of this wave: 64 lanes x 16 B from (uniform 64-bit base + per-lane 32-bit offset) into 1 KiB of LDS at a uniform address.
struct F6Piece { const char* src; int64_t kstride; int lds; };

template <int EPI, class G>
__global__ __launch_bounds__(G::NTHR) void gemm_f16f6_kernel(F6P p) {
    constexpr int WN = G::WN, TM = G::TM, TN = G::TN, NST = G::NST, BM = G::BM, BN = G::BN, NW = G::NW, SLOT = G::SLOT;
    // DMA schedule of a slot.  The pieces are dealt so that every wave's share is the same straight-line code: H pieces (16 rows x 64 B,
    // XOR-swizzled source chunks) PAH / NW rounds of A and PBH / NW rounds of B (+ a partial round for the waves below HB_REM), then the
    // 24 "linear" pieces (A_FH, A_FL, B_FH, B_FL copies and the two scale pieces) NLIN / NW rounds.  Waves below HB_REM issue one piece more.
    constexpr int NLIN = 2 * G::PAF + 2 * G::PBF + 2;
    static_assert(G::PAH % NW == 0 && NLIN % NW == 0, "the A_H and the linear pieces must split evenly over the waves");
    constexpr int HA_R = G::PAH / NW, HB_R = G::PBH / NW, HB_REM = G::PBH % NW, LIN_R = NLIN / NW;
    constexpr int CNT_LO = HA_R + HB_R + LIN_R;                     // pieces per slot of a wave >= HB_REM (the others: one more)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);        // scalar: every piece address below is SGPR base + VGPR lane offset
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_n = ((EPI == F6_EPI_PLANES ? p.Np : p.N) + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const bool extra = wid < HB_REM;
    // per-lane source offsets: H piece = rows (lane >> 2) of the piece, chunk (lane & 3) ^ ((row >> 2) & 3) -- a piece starts at a multiple of 16 rows
    const unsigned hoff = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    const unsigned loff = (unsigned)lane * 16u;

    // Persistent tile walk.  The NEXT tile's first NST - 1 slots are issued BEFORE the current tile's epilogue stores (non-staged
    // epilogues): the ring is idle by then, and the DMA latency hides behind the store issue (the stores are issue-bound, ~10 us per tile).
    int z = 0, m0 = 0, n0 = 0;
    int64_t rowA = 0, rowB = 0;
    F6Piece pa[HA_R], pb[HB_R + 1], pl[LIN_R];
#ifndef CTI_F6_PREFETCH
#define CTI_F6_PREFETCH 0        // > 0: every wave touches (one dword per 128-B line, 64 lines) part of the K block issued PREFETCH blocks later.
#endif                           // MEASURED AND REJECTED (2.0 -> 2.3 ms at distances 2, 3, 5): the extra requests cost more than the earlier misses save
    constexpr int NPF = CTI_F6_PREFETCH > 0 ? 1 : 0;               // prefetch loads per wave and slot (they count in vmcnt like the DMA pieces)
    const char* pf_src = nullptr; int64_t pf_kstride = 0;
    int pf_sink = 0;                                                // destination register of every prefetch load: stays reserved to the end
    auto setup_tile = [&](int vt) {
        int tm, tn;
        tile_coords(vt, p.total_tiles, tiles_m, tiles_n, z, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
        rowA = (int64_t)z * p.rA + m0; rowB = (int64_t)z * p.rB + n0;
        // this wave's pieces (uniform): source base at K block 0, byte stride per K block, LDS offset inside the slot
#pragma unroll
        for (int u = 0; u < HA_R; ++u) {
            const int q = wid + u * NW;
            pa[u].src = p.AH + (rowA + q * 16) * 64; pa[u].kstride = p.pA * 64; pa[u].lds = G::OFF_AH + q * 1024;
        }
#pragma unroll
        for (int u = 0; u < HB_R + 1; ++u) {
            const int q = wid + u * NW;                                 // u == HB_R: only the waves below HB_REM
            pb[u].src = p.BH + (rowB + q * 16) * 64; pb[u].kstride = p.pB * 64; pb[u].lds = G::OFF_BH + q * 1024;
        }
#pragma unroll
        for (int u = 0; u < LIN_R; ++u) {
            int g = wid + u * NW;                                       // index among the linear pieces: [A_FH | A_FL | A_S | B_FH | B_FL | B_S]
            const char* base; int64_t ks; int lds;
            if (g < G::PAF)                         { base = p.AFH + rowA * 24; ks = p.pA * 24; lds = G::OFF_AFH; }
            else if ((g -= G::PAF) < G::PAF)        { base = p.AFL + rowA * 24; ks = p.pA * 24; lds = G::OFF_AFL; }
            else if ((g -= G::PAF) < 1)             { base = p.AS + rowA * 2;   ks = p.pAS * 2; lds = G::OFF_AS; }
            else if ((g -= 1) < G::PBF)             { base = p.BFH + rowB * 24; ks = p.pB * 24; lds = G::OFF_BFH; }
            else if ((g -= G::PBF) < G::PBF)        { base = p.BFL + rowB * 24; ks = p.pB * 24; lds = G::OFF_BFL; }
            else                                    { g -= G::PBF; base = p.BS + rowB * 2; ks = p.pBS * 2; lds = G::OFF_BS; }
            pl[u].src = base + g * 1024; pl[u].kstride = ks; pl[u].lds = lds + g * 1024;
        }
        if (NPF) {
            // L2 prefetch experiment (off).  The ring holds two K blocks in flight (104 KiB); the idea was that a touch of the lines a LATER
            // block will DMA moves their miss out of the DMA -- more bytes in flight without LDS to land them in.  It made the kernel slower.  wave: 0,1 B_H halves (96 lines), 2 B_FH, 3 B_FL, 4,5 A_H halves, 6 A_FH, 7 A_FL
            const int64_t lo = (int64_t)lane * 128;
            switch (wid) {
                case 0: pf_src = p.BH + rowB * 64 + lo; pf_kstride = p.pB * 64; break;
                case 1: pf_src = p.BH + rowB * 64 + (lane < 32 ? 8192 + lo : lo); pf_kstride = p.pB * 64; break;
                case 2: pf_src = p.BFH + rowB * 24 + (lane < 36 ? lo : 0); pf_kstride = p.pB * 24; break;
                case 3: pf_src = p.BFL + rowB * 24 + (lane < 36 ? lo : 0); pf_kstride = p.pB * 24; break;
                case 4: pf_src = p.AH + rowA * 64 + lo; pf_kstride = p.pA * 64; break;
                case 5: pf_src = p.AH + rowA * 64 + 8192 + lo; pf_kstride = p.pA * 64; break;
                case 6: pf_src = p.AFH + rowA * 24 + (lane < 48 ? lo : 0); pf_kstride = p.pA * 24; break;
                default: pf_src = p.AFL + rowA * 24 + (lane < 48 ? lo : 0); pf_kstride = p.pA * 24; break;
            }
        }
    };
    auto issue_slot = [&](int pos, int64_t kb) {
        char* slot = smem + pos * SLOT;
#pragma unroll
        for (int u = 0; u < HA_R; ++u) dma16(pa[u].src + kb * pa[u].kstride + hoff, slot + pa[u].lds);
#pragma unroll
        for (int u = 0; u < HB_R; ++u) dma16<CTI_F6_NT_B ? 2 : 0>(pb[u].src + kb * pb[u].kstride + hoff, slot + pb[u].lds);
        if (HB_REM && extra) dma16<CTI_F6_NT_B ? 2 : 0>(pb[HB_R].src + kb * pb[HB_R].kstride + hoff, slot + pb[HB_R].lds);
#pragma unroll
        for (int u = 0; u < LIN_R; ++u) dma16(pl[u].src + kb * pl[u].kstride + loff, slot + pl[u].lds);
        if (NPF) {                                                  // exactly one load per slot, whatever kb (the counted waits rely on it)
            const int64_t kp = kb + CTI_F6_PREFETCH < p.Kb ? kb + CTI_F6_PREFETCH : p.Kb - 1;
            const char* a = pf_src + kp * pf_kstride;
            asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(a) : "memory");
        }
    };
    const int nkb = p.Kb;
    auto prologue = [&]() {
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) if (i < nkb) issue_slot(i, i);
    };
    constexpr bool HOIST = EPI != F6_EPI_PLANES;                    // the planes epilogue stages through the ring's LDS

    int vtile = blockIdx.x;
    if (vtile >= p.total_tiles) return;
    // De-synchronise the CUs.  Every workgroup walks tiles of equal cost, so all 256 would reach their epilogues together: 50 MB of stores
    // per round hit HBM at once (write-bound, ~12 us) and nothing is stored in between.  Spreading the START over one tile time keeps the
    // phases apart for the whole launch: the stores become a steady stream beside the other CUs' main loops.
    if (p.desync_ticks > 0) {
        const unsigned target = (((unsigned)blockIdx.x * 157u) & 255u) * (unsigned)p.desync_ticks >> 8;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((unsigned)(__builtin_amdgcn_s_memrealtime() - t0) < target) __builtin_amdgcn_s_sleep(16);
    }
    setup_tile(vtile);
    prologue();
    for (;;) {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // per-lane fragment addresses inside a slot
    int aH[TM], aF[TM], aS[TM], bH[TN], bF[TN], bS[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + r;
        aH[i] = G::OFF_AH + row * 64; aF[i] = (h ? G::OFF_AFL : G::OFF_AFH) + row * 24; aS[i] = G::OFF_AS + row * 2 + h;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + r;
        bH[j] = G::OFF_BH + row * 64; bF[j] = (h ? G::OFF_BFH : G::OFF_BFL) + row * 24; bS[j] = G::OFF_BS + row * 2 + (1 - h);
    }
    const int sw = (r >> 2) & 3;                                    // (row >> 2) & 3: rows of a tile start at multiples of 32
    const int c0 = ((0 + h) ^ sw) << 4, c1 = ((2 + h) ^ sw) << 4;   // swizzled chunk offsets of k-steps 0 and 1

    // K loop.  "sync(b)": this wave's pieces of block b have landed (counted vmcnt), raw barrier (every wave's have, and block b - 1 is
    // free), refill: block b + NST - 1 is issued into the freed slot.
    // STAGGER: waves w and w + NW/2 share a SIMD and run the same program; released by the same barrier they would both read their
    // fragments first (LDS latency exposed, matrix pipe idle) and then both queue MFMAs.  The upper half therefore reads block b BEFORE
    // sync(b + 1) and issues its MFMAs AFTER it: one SIMD partner computes while the other reads.  Its reads are complete (lgkmcnt(0))
    // before the barrier, so the slot may be recycled behind it; fragments never live across a loop back-edge.
#ifndef CTI_F6_STAGGER
#define CTI_F6_STAGGER 1
#endif
    auto sync_only = [&](int b) {
        const int rem = nkb - 1 - b;                                // blocks issued after block b so far: min(NST - 2, rem)
#if CTI_F6_ABL & 32
        if (b < NST - 1 && vtile != (int)blockIdx.x) { __builtin_amdgcn_s_barrier(); return; }     // ablation (UNSAFE): no vmcnt wait on the hoisted blocks
#endif
        if (rem >= NST - 2) { if (extra) wait_vm<(NST - 2) * (CNT_LO + 1 + NPF)>(); else wait_vm<(NST - 2) * (CNT_LO + NPF)>(); }
        else if (NST >= 4 && rem == 1) { if (extra) wait_vm<CNT_LO + 1 + NPF>(); else wait_vm<CNT_LO + NPF>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    };
    auto refill = [&](int b, int pos) {                             // pos = ring position of block b, whose barrier has been passed
        if (b + NST - 1 < nkb && !(CTI_F6_ABL & 1)) issue_slot(pos == 0 ? NST - 1 : pos - 1, b + NST - 1);
    };
#ifndef CTI_F6_LAG_DMA_LATE
#define CTI_F6_LAG_DMA_LATE 1    // the lagging waves issue their share of the refill BEHIND their MFMAs: right after a barrier the lead waves
#endif                           // issue DMA (no matrix work yet) while the lagging ones feed the matrix pipe, then the roles swap
#define CTI_F6_READ_FRAGS(s)                                                                                                          \
    f16x8 a16[TM][2], b16[TN][2]; i32x8 a6[TM], b6[TN]; int sa[TM], sb[TN];                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                                  \
        a16[i][0] = *reinterpret_cast<const f16x8*>((s) + aH[i] + c0);                                                                \
        a16[i][1] = *reinterpret_cast<const f16x8*>((s) + aH[i] + c1);                                                                \
        const u32x2 f0 = *reinterpret_cast<const u32x2*>((s) + aF[i]), f1 = *reinterpret_cast<const u32x2*>((s) + aF[i] + 8), f2 = *reinterpret_cast<const u32x2*>((s) + aF[i] + 16); \
        a6[i][0] = f0.x; a6[i][1] = f0.y; a6[i][2] = f1.x; a6[i][3] = f1.y; a6[i][4] = f2.x; a6[i][5] = f2.y; a6[i][6] = 0; a6[i][7] = 0; \
        sa[i] = *reinterpret_cast<const int*>((s) + (aS[i] & ~3)) >> ((aS[i] & 3) * 8);   /* the MFMA takes byte 0 of the scale register */ \
    }                                                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                                                  \
        b16[j][0] = *reinterpret_cast<const f16x8*>((s) + bH[j] + c0);                                                                \
        b16[j][1] = *reinterpret_cast<const f16x8*>((s) + bH[j] + c1);                                                                \
        const u32x2 f0 = *reinterpret_cast<const u32x2*>((s) + bF[j]), f1 = *reinterpret_cast<const u32x2*>((s) + bF[j] + 8), f2 = *reinterpret_cast<const u32x2*>((s) + bF[j] + 16); \
        b6[j][0] = f0.x; b6[j][1] = f0.y; b6[j][2] = f1.x; b6[j][3] = f1.y; b6[j][4] = f2.x; b6[j][5] = f2.y; b6[j][6] = 0; b6[j][7] = 0; \
        sb[j] = *reinterpret_cast<const int*>((s) + (bS[j] & ~3)) >> ((bS[j] & 3) * 8);                                               \
    }
#define CTI_F6_MFMAS()                                                                                                                \
    if (!(CTI_F6_ABL & 2) || p.Kb < 0) {         /* ablation: never true, the fragments stay live, the MFMAs do not issue */          \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                                \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                                          \
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6[i], b6[j], acc[i][j], 2, 2, 0, sa[i], 0, sb[j]);       \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[i][0], b16[j][0], acc[i][j], 0, 0, 0);                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[i][1], b16[j][1], acc[i][j], 0, 0, 0);                         \
            }                                                                                                                         \
    }
    const bool lag = CTI_F6_STAGGER && wid >= NW / 2;
    if (!lag) {
        int pos = 0;
        for (int kb = 0; kb < nkb; ++kb) {
            sync_only(kb); refill(kb, pos);
            const char* s = smem + pos * SLOT;
            CTI_F6_READ_FRAGS(s)
            CTI_F6_MFMAS()
            pos = pos == NST - 1 ? 0 : pos + 1;
        }
        __syncthreads();                          // every wave is done reading the ring before the epilogue / the next tile's DMA reuses it
    } else {
        int pos = 0;
        sync_only(0); refill(0, 0);
        for (int kb = 0; kb < nkb; ++kb) {
            const char* s = smem + pos * SLOT;
            CTI_F6_READ_FRAGS(s)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads complete before the barrier that lets the slot be recycled
            pos = pos == NST - 1 ? 0 : pos + 1;
            if (kb + 1 < nkb) { sync_only(kb + 1); if (!CTI_F6_LAG_DMA_LATE) refill(kb + 1, pos); }
            else __syncthreads();                 // pairs with the lead waves' closing barrier
            CTI_F6_MFMAS()
            if (CTI_F6_LAG_DMA_LATE && kb + 1 < nkb) { __builtin_amdgcn_sched_barrier(0); refill(kb + 1, pos); }
        }
    }
#undef CTI_F6_READ_FRAGS
#undef CTI_F6_MFMAS

    float* C = p.C + (int64_t)z * p.sC;
    const int cm0 = m0, cn0 = n0;                 // the finished tile's origin (setup_tile moves on to the next one)
    const int next = vtile + (int)gridDim.x;
    const bool have_next = next < p.total_tiles;
    if (HOIST && have_next) { setup_tile(next); prologue(); }
    if (EPI == F6_EPI_INTERLEAVE2) {
        // GEMM rows (2m, 2m+1) are the two glimpses of one (v,q) row: registers e, e+1 (e even) of a lane are the adjacent floats
        // out[b, vq, a, 0:2] of column a = n.  Neighbouring lanes (columns n, n+1) trade halves through a DPP quad swap so that the even lane
        // stores out[vq, n:n+2, 0:2] and the odd lane out[vq+1, n-1:n+1, 0:2]: ONE 16-B store per four registers instead of two 8-B ones --
        // the epilogue is store-ISSUE bound (cdna_hip_programming.md T21), so this halves it.
        const bool odd = lane & 1;
        auto swap1 = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); };
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = cn0 + (wn * TN + j) * 32 + r - (odd ? 1 : 0);      // even column: this lane's 16 B start here
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float r0 = acc[i][j][e], r1 = acc[i][j][e + 1], r2 = acc[i][j][e + 2], r3 = acc[i][j][e + 3];
                    const float t0 = swap1(odd ? r0 : r2), t1 = swap1(odd ? r1 : r3);
                    const int m = cm0 + (wm * TM + i) * 32 + 8 * (e >> 2) + 4 * h + (odd ? 2 : 0);      // even GEMM row = (vq, g = 0)
                    if (m >= p.M || col >= p.N || ((CTI_F6_ABL & 8) && r0 != 12345.f)) continue;
                    float* dst = C + (int64_t)(m >> 1) * p.ldc_m + (int64_t)col * 2;
#if CTI_F6_ABL & 16
                    dst = p.C + ((((int64_t)(m >> 1) * p.ldc_m + (int64_t)col * 2) & 0x3ffff) + (blockIdx.x & 7) * 0x40000);   // ablation: every store lands in an L2-resident 8 MiB
#endif
#ifndef CTI_F6_WIDE_STORES
#define CTI_F6_WIDE_STORES 1
#endif
                    if (!CTI_F6_WIDE_STORES) {
                        typedef float f32x2n __attribute__((ext_vector_type(2)));
                        // un-widened form (8-B stores): even lane row m, odd lane row m too -- use the lane's own registers
                        const int mo = cm0 + (wm * TM + i) * 32 + 8 * (e >> 2) + 4 * h, no = cn0 + (wn * TN + j) * 32 + r;
                        if (no < p.N) {
                            f32x2n w0; w0[0] = r0; w0[1] = r1; f32x2n w1; w1[0] = r2; w1[1] = r3;
                            if (mo < p.M) *reinterpret_cast<f32x2n*>(C + (int64_t)(mo >> 1) * p.ldc_m + (int64_t)no * 2) = w0;
                            if (mo + 2 < p.M) *reinterpret_cast<f32x2n*>(C + (int64_t)((mo >> 1) + 1) * p.ldc_m + (int64_t)no * 2) = w1;
                        }
                        continue;
                    }
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef CTI_F6_STORE_SC1
#define CTI_F6_STORE_SC1 0       // 1: write-through stores that do NOT keep the line in the XCD's L2 (the 3.2 GB output stream would otherwise
#endif                           // evict the operand tiles the LDS-DMA re-reads from L2)
                    if (col + 1 < p.N) {
                        f32x4 v4; v4[0] = odd ? t0 : r0; v4[1] = odd ? t1 : r1; v4[2] = odd ? r2 : t0; v4[3] = odd ? r3 : t1;
                        if (CTI_F6_STORE_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v4) : "memory");
                        else *reinterpret_cast<f32x4*>(dst) = v4;
                    } else {
                        f32x2 v2; v2[0] = odd ? t0 : r0; v2[1] = odd ? t1 : r1;
                        if (CTI_F6_STORE_SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v2) : "memory");
                        else *reinterpret_cast<f32x2*>(dst) = v2;
                    }
                }
        }
    } else if (EPI == F6_EPI_PLANES) {
        // Per 32-column tile j the wave parks its 64 x 32 block in a private LDS patch ([row][36 floats]: conflict-free b128 row reads), then
        // every lane encodes ONE (row, block) item; bias / scale / ReLU applied on the way in.  Same wave writes and reads: in-order LDS.
        float* stg = reinterpret_cast<float*>(smem) + wid * (TM * 32 * 36);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = cn0 + (wn * TN + j) * 32, n = nb + r;
            const bool real = n < p.N;
            const float sc = (real && p.scale) ? p.scale[n / p.scale_div] : 1.f;
            const float bi = (real && p.bias) ? p.bias[n] : 0.f;
            if (j) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous tile's reads are done before overwriting
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    float x = acc[i][j][e] * sc + bi;
                    if (p.relu) x = fmaxf(x, 0.f);
                    stg[row * 36 + r] = real ? x : 0.f;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_assert(TM * 32 == 64, "one (row, block) item per lane");
            const int m = cm0 + wm * TM * 32 + lane;
            if (m < p.M && nb < p.Np) f6_encode_row32_lds(stg + lane * 36, p.P, f6_prow(p.P, m), nb >> 5);
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = cn0 + (wn * TN + j) * 32 + r;
            if (n >= p.N) continue;
            const float sc = p.scale ? p.scale[n / p.scale_div] : 1.f;
            const float bi = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = cm0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (m < p.M) {
                        float x = acc[i][j][e] * sc + bi;
                        if (p.relu) x = fmaxf(x, 0.f);
                        C[(int64_t)(m / p.gdiv) * p.ldc_m + (m % p.gdiv) + (int64_t)n * p.ldc_n] = x;
                    }
                }
        }
    }
    if (!have_next) break;
    if (!HOIST) {
        __syncthreads();                          // (planes epilogue) the LDS patches are free again before the next tile's DMA
        setup_tile(next); prologue();
    }
    vtile = next;
    }                                             // persistent tile loop
    if (NPF) asm volatile("s_waitcnt vmcnt(0)\n; prefetch sink %0" ::"v"(pf_sink) : "memory");
}

using GeoF6 = GeoF<4, 2, 2, 3, 3>;                // 256 x 192, 3 slots of 52 KiB

template <int EPI>
int launch_f6(const F6P& p0, long long nb, int ncols, hipStream_t st) {
    using G = GeoF6;
    auto kern = gemm_f16f6_kernel<EPI, G>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail((int)e, "gemm_nt_f16f6: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    const long long total = nb * ((p0.M + G::BM - 1) / G::BM) * ((ncols + G::BN - 1) / G::BN);
    if (total > 0x7fffffffLL) return fail(CTI_E_SHAPE, "gemm_nt_f16f6: %lld tiles exceed the grid", total);
    F6P p = p0;
    p.total_tiles = (int)total;
    static thread_local int n_cu = 0;
    if (n_cu == 0) { (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    long long grid = n_cu;
    if (grid > total) grid = total;
#ifndef CTI_F6_DESYNC
#define CTI_F6_DESYNC 100        // percent of one estimated tile time
#endif
    // one tile ~ 2 * BM * BN * K flops at ~2 TFLOP/s per CU (the measured full-chip rate of this kernel / 256)
    p.desync_ticks = total >= 2 * grid ? (int)(2.0 * G::BM * G::BN * p.Kb * 32 / 2.0e12 * 1.0e8 * CTI_F6_DESYNC / 100.0) : 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NTHR), G::LDS, st, p);
    return launch_status("gemm_nt_f16f6");
}

}  // namespace

int quantize_f16f6(const float* x, int64_t ld, int64_t rows, int K, const F6Planes& p, hipStream_t st) {
    if (rows <= 0 || K <= 0) return fail(CTI_E_SHAPE, "quantize_f16f6: rows=%lld K=%d", (long long)rows, K);
    const int64_t bx = (rows + 255) / 256;
    if (bx > 0x7fffffffLL || p.Kb > 65535) return fail(CTI_E_SHAPE, "quantize_f16f6: rows=%lld K=%d exceed the grid", (long long)rows, K);
    hipLaunchKernelGGL(quantize_f16f6_kernel, dim3((unsigned)bx, (unsigned)p.Kb), dim3(256), 0, st, x, ld, rows, K, p);
    return launch_status("quantize_f16f6");
}

int gemm_nt_f16f6(const F6GemmArgs& a, hipStream_t st) {
    if (a.A.Kb != a.B.Kb || a.A.Kb <= 0) return fail(CTI_E_SHAPE, "gemm_nt_f16f6: K blocks %d vs %d", a.A.Kb, a.B.Kb);
    if (a.M <= 0 || a.N <= 0 || a.nb <= 0) return fail(CTI_E_SHAPE, "gemm_nt_f16f6: M=%d N=%d nb=%d", a.M, a.N, a.nb);
    if ((a.A.rows_alloc | a.A.rows_allocS | a.B.rows_alloc | a.B.rows_allocS) & 7) return fail(CTI_E_ALIGN, "gemm_nt_f16f6: plane row counts must be multiples of 8");
    if (a.nb > 1 && ((a.rA | a.rB) & 7)) return fail(CTI_E_ALIGN, "gemm_nt_f16f6: batch strides rA=%lld rB=%lld must be multiples of 8 rows", (long long)a.rA, (long long)a.rB);
    F6P p{};
    p.AH = reinterpret_cast<const char*>(a.A.H); p.AFH = reinterpret_cast<const char*>(a.A.FH); p.AFL = reinterpret_cast<const char*>(a.A.FL); p.AS = reinterpret_cast<const char*>(a.A.S);
    p.BH = reinterpret_cast<const char*>(a.B.H); p.BFH = reinterpret_cast<const char*>(a.B.FH); p.BFL = reinterpret_cast<const char*>(a.B.FL); p.BS = reinterpret_cast<const char*>(a.B.S);
    p.pA = a.A.rows_alloc; p.pAS = a.A.rows_allocS; p.pB = a.B.rows_alloc; p.pBS = a.B.rows_allocS;
    p.rA = a.rA; p.rB = a.rB; p.M = a.M; p.N = a.N; p.Kb = a.A.Kb;
    p.C = a.C; p.ldc_m = a.ldc_m; p.ldc_n = a.ldc_n; p.sC = a.sC; p.gdiv = a.gdiv > 0 ? a.gdiv : 1;
    p.scale = a.scale; p.scale_div = a.scale_div > 0 ? a.scale_div : 1; p.bias = a.bias; p.relu = a.relu;
    p.P = a.P; p.Np = a.Np;
    switch (a.epi) {
        case 0: p.gdiv = 1; return launch_f6<F6_EPI_F32>(p, a.nb, a.N, st);
        case 3:
            if (p.gdiv == 2 && a.ldc_n == 2) return launch_f6<F6_EPI_INTERLEAVE2>(p, a.nb, a.N, st);
            return launch_f6<F6_EPI_INTERLEAVE>(p, a.nb, a.N, st);
        case 4:
            if (a.nb != 1) return fail(CTI_E_UNSUPPORTED, "gemm_nt_f16f6: planes output needs nb = 1");
            if (a.Np % 32 != 0 || a.Np < a.N) return fail(CTI_E_SHAPE, "gemm_nt_f16f6: Np=%d (N=%d) must be a multiple of 32", a.Np, a.N);
            return launch_f6<F6_EPI_PLANES>(p, a.nb, a.Np, st);
        default: return fail(CTI_E_UNSUPPORTED, "gemm_nt_f16f6: epi=%d", a.epi);
    }
}

}  // namespace cti

using namespace cti;

extern "C" size_t cti_f16f6_planes_bytes(int64_t rows, int K, int64_t batch_rows) {
    return rows > 0 && K > 0 && batch_rows >= 0 ? f6_planes_bytes(rows, K, batch_rows) : 0;
}

extern "C" int cti_quantize_f16f6(const float* x, int64_t ld, int64_t rows, int K, int64_t batch_rows, void* planes, size_t planes_bytes, void* stream) {
    CTI_REQUIRE_PTR(x); CTI_REQUIRE_PTR(planes);
    CTI_REQUIRE(rows > 0 && K > 0 && ld >= K && batch_rows >= 0, CTI_E_SHAPE, "cti_quantize_f16f6: rows=%lld K=%d ld=%lld batch_rows=%lld", (long long)rows, K, (long long)ld, (long long)batch_rows);
    CTI_REQUIRE(planes_bytes >= f6_planes_bytes(rows, K, batch_rows), CTI_E_WORKSPACE, "cti_quantize_f16f6: block %zu < %zu", planes_bytes, f6_planes_bytes(rows, K, batch_rows));
    CTI_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 255) == 0, CTI_E_ALIGN, "cti_quantize_f16f6: the plane block must be 256-B aligned");
    hipError_t e = hipMemsetAsync(planes, 0, f6_planes_bytes(rows, K, batch_rows), as_stream(stream));       // slack / padding rows: defined scales, zero codes
    if (e != hipSuccess) return fail((int)e, "cti_quantize_f16f6: hipMemsetAsync: %s", hipGetErrorString(e));
    return quantize_f16f6(x, ld, rows, K, f6_carve(planes, rows, K, batch_rows), as_stream(stream));
}

extern "C" int cti_gemm_nt_f16f6(const void* A_planes, int64_t rowsA_total, int64_t batch_rowsA, const void* B_planes, int64_t rowsB_total, int64_t batch_rowsB,
                                 float* C, int64_t ldc_m, int64_t ldc_n, int64_t sC, int gdiv, int nb, int M, int N, int K, const float* scale, int scale_div,
                                 const float* bias, int act, void* stream) {
    CTI_REQUIRE_PTR(A_planes); CTI_REQUIRE_PTR(B_planes); CTI_REQUIRE_PTR(C);
    CTI_REQUIRE(M > 0 && N > 0 && K > 0 && nb > 0 && gdiv > 0, CTI_E_SHAPE, "cti_gemm_nt_f16f6: M=%d N=%d K=%d nb=%d gdiv=%d", M, N, K, nb, gdiv);
    CTI_REQUIRE(nb == 1 ? (M <= rowsA_total && N <= rowsB_total) : (M <= batch_rowsA && N <= batch_rowsB && (int64_t)nb * batch_rowsA <= rowsA_total && (int64_t)nb * batch_rowsB <= rowsB_total),
                CTI_E_SHAPE, "cti_gemm_nt_f16f6: batches run past the operand rows");
    CTI_REQUIRE(act == CTI_ACT_NONE || act == CTI_ACT_RELU, CTI_E_UNSUPPORTED, "cti_gemm_nt_f16f6: act=%d", act);
    F6GemmArgs g{};
    g.A = f6_carve(const_cast<void*>(A_planes), rowsA_total, K, batch_rowsA); g.B = f6_carve(const_cast<void*>(B_planes), rowsB_total, K, batch_rowsB);
    g.rA = g.A.rstride; g.rB = g.B.rstride; g.nb = nb; g.M = M; g.N = N;
    g.epi = gdiv > 1 ? 3 : 0; g.C = C; g.ldc_m = ldc_m; g.ldc_n = ldc_n; g.sC = sC; g.gdiv = gdiv;
    g.scale = scale; g.scale_div = scale_div; g.bias = bias; g.relu = act == CTI_ACT_RELU;
    return gemm_nt_f16f6(g, as_stream(stream));
}
