// cti_gemm_bf16x3.hip -- fp32-grade NT GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16).
//
// gfx950 has no TF32/xf32; its exact-fp32 MFMA runs at 1/16 of the bf16 rate.  Here every fp32 operand x is split
// once into two bf16 planes, hi = bf16(x) and lo = bf16(x - hi) (|x - hi - lo| <= 2^-17 |x|), and each product is
// issued as three bf16 MFMAs into one fp32 accumulator:  a*b ~= ah*bh + ah*bl + al*bh  (the dropped al*bl term is
// <= 2^-16 relative).  Measured against the float64 oracle at the BASELINE config-1 shapes the whole TCNet.forward
// stays at 1e-5 normalised max error (tolerance 1e-4), at 3/16 of the exact-fp32 MFMA issue cost.
//
// PLANE LAYOUT (chunk-major).  A plane of a (rows x K) operand is stored as [K/16 chunks][rows_alloc][16 elements]:
// element (row, k) lives at (k >> 4) * pitch + row * 16 + (k & 15), pitch = rows_alloc * 16.  One 16-deep K slice of a
// GEMM tile (R rows) is therefore ONE contiguous R*32-byte run, every LDS-DMA wave-instruction reads 1 KiB contiguous
// (32 rows), and neighbouring rows are 32 B apart (no power-of-two row pitch: no L2-channel camping).  The kernel was measured to be
// bound by the L2 -> LDS DMA path (12 TB/s chip-wide with 64-B row segments, 16 TB/s with contiguous KiB), so layout and
// tile size are chosen to minimise DMA requests and bytes; every producer (split kernel, GEMM epilogue, M build)
// writes this layout directly.  K tails (K..Kp-1, Kp = K rounded up to 32) are zero-filled by the producer; rows beyond
// the last valid one (up to PLANE_SLACK_ROWS) only feed discarded outputs.
//
// KERNEL.  Tile (WM*TM*32) x (WN*TN*32) per workgroup of WM*WN consumer waves (+ LW loader waves); an NST-slot LDS ring
// of 16-deep K slices filled by LDS-DMA (global_load_lds_dwordx4: 16 B per lane straight into LDS) with COUNTED vmcnt;
// SPB slices are consumed per raw s_barrier, NST - SPB slices stay in flight behind the MFMAs (the 256 x 256 tile keeps
// three 32-KiB slices in flight: one 64-KiB K-step in flight was measured latency-bound).  The LDS image is lane-linear
// ([row][2 x 16 B]) and made bank-conflict-free for ds_read_b128 by permuting the SOURCE chunk
// (c' = c ^ ((row >> 3) & 1)) and applying the same XOR on the read.
#include "cti_common.h"
#include "cti_f16f6.h"
#include <cstdlib>
#include <type_traits>

#ifndef CTI_PIPE
#define CTI_PIPE 0
#endif
#ifndef CTI_ABL                 // timing-only ablations (wrong results): 1 no refill DMA, 2 no MFMA, 4 no epilogue stores
#define CTI_ABL 0
#endif

namespace cti {

namespace {

constexpr int BK = 16;                                    // K depth of one ring slot = one MFMA 32x32x16 step
constexpr int KPAD = 32;                                  // planes are zero-filled up to a multiple of 32 in K
constexpr int ROW_BYTES = BK * 2;                         // one slot row of one plane: 32 B = 2 chunks of 16 B
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x)); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ uint4 pack8(const unsigned short* h) {
    return make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
}

// ---- split: fp32 [rows, K] (row stride ld) -> chunk-major hi/lo planes --------------------------------------------
// grid.y = 32-wide K group; thread = (row, 16-B piece c of the group): piece c belongs to chunk 2*group + (c >> 1).
// Reads are 32 B per thread, 128 B contiguous per row; stores 16 B per thread, 32 B contiguous per row and chunk.
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int K, int Kp,
                                                    unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                    int64_t pitch) {
#ifndef CTI_SPLIT_KFAST
#define CTI_SPLIT_KFAST 1        // K group on the fast grid axis: workgroups dispatched together read the SAME rows (whole 1.2-KB rows of `a` instead of
#endif                           // one 128-B piece of every row per pass)
#if CTI_SPLIT_KFAST
    const int64_t idx = ((int64_t)blockIdx.y * gridDim.z + blockIdx.z) * 256 + threadIdx.x;
    const int kg = blockIdx.x;
#else
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;     // (row, 16-B piece) inside K chunk blockIdx.y
    const int kg = blockIdx.y;
#endif
    const int c = (int)(idx & 3);
    const int64_t row = idx >> 2;
    if (row >= rows) return;
    const int k0 = kg * 32 + c * 8;
    const float* src = x + row * ld + k0;
    float v[8];
    if (k0 + 8 <= K && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k0 + j < K) ? src[j] : 0.f;
    }
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = bf16_bits(v[j]);
        l[j] = bf16_bits(v[j] - bf16_to_f32(h[j]));
    }
    const int64_t o = (int64_t)(k0 >> 4) * pitch + row * 16 + (k0 & 15);
    *reinterpret_cast<uint4*>(hi + o) = pack8(h);
    if (lo) *reinterpret_cast<uint4*>(lo + o) = pack8(l);           // lo == NULL: the plain-bf16 mode's operands (one product per pair reads the hi plane only)
}

// bf16 rows -> chunk-major planes (round 5: a producer's bf16 output as the next product's planes operand): the hi plane takes the bits as they are, the lo
// plane (three-product callers) zeros.  Same thread map as split_kernel; K % 8 == 0, 16-B aligned rows.
__global__ __launch_bounds__(256) void split16_kernel(const unsigned short* __restrict__ x, int64_t ld, int64_t rows, int K, unsigned short* __restrict__ hi,
                                                      unsigned short* __restrict__ lo, int64_t pitch) {
    const int64_t idx = ((int64_t)blockIdx.y * gridDim.z + blockIdx.z) * 256 + threadIdx.x;
    const int kg = blockIdx.x;
    const int c = (int)(idx & 3);
    const int64_t row = idx >> 2;
    if (row >= rows) return;
    const int k0 = kg * 32 + c * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (k0 + 8 <= K) v = *reinterpret_cast<const uint4*>(x + row * ld + k0);
    const int64_t o = (int64_t)(k0 >> 4) * pitch + row * 16 + (k0 & 15);
    *reinterpret_cast<uint4*>(hi + o) = v;
    if (lo) *reinterpret_cast<uint4*>(lo + o) = make_uint4(0u, 0u, 0u, 0u);
}

// Transposing split: x is (M x n) row-major; the planes hold x^T, i.e. rows = the n columns of x, depth = M (zero-filled up to Mp).
// A thread owns one column: its 16 loads (one per row of the chunk) are coalesced across the workgroup, and its 16 values are one
// contiguous 32-B run of each plane.  This is the weight-gradient operand layout (dW = dz^T x contracts over the ROW axis) without
// a transposed fp32 copy.
__global__ __launch_bounds__(256) void split_t_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int n, unsigned short* __restrict__ hi,
                                                      unsigned short* __restrict__ lo, int64_t pitch) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= n) return;
    const int64_t m0 = (int64_t)blockIdx.y * 16;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (m0 + j < M) ? x[(m0 + j) * ld + col] : 0.f;
    unsigned short h[16], l[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { h[j] = bf16_bits(v[j]); l[j] = bf16_bits(v[j] - bf16_to_f32(h[j])); }
    const int64_t o = (int64_t)blockIdx.y * pitch + (int64_t)col * 16;
    *reinterpret_cast<uint4*>(hi + o) = pack8(h);     *reinterpret_cast<uint4*>(hi + o + 8) = pack8(h + 8);
    *reinterpret_cast<uint4*>(lo + o) = pack8(l);     *reinterpret_cast<uint4*>(lo + o + 8) = pack8(l + 8);
}

struct PlaneGemmP {
    const unsigned short* Ah; const unsigned short* Al; const unsigned short* Bh; const unsigned short* Bl;
    float* C;
    int64_t pitchA, pitchB;                    // chunk pitches (elements) of the operand planes
    int64_t ldc_m, ldc_n;
    int64_t rA1, rA2, rB1, rB2;                // batch strides of the operands, in ROWS
    int64_t kc2;                               // split-K: batch b2 starts kc2 * b2 K-chunks (of 16) into both operands
    int64_t sC1, sC2;                          // batch strides of C (elements) / of the output planes (rows)
    int nb2;
    int M, N, Kp;
    int total_tiles;
    const float* Af; int64_t ldaf; int Kreal;  // AF32 mode: the A operand is the fp32 matrix itself (rows x Kreal, row stride ldaf)
    const float* scale; int scale_div; const float* bias; int relu;
    int64_t scale_bs, bias_bs;                 // per-batch (b1) strides of scale / bias
    // EPI_PLANES: the result is written as chunk-major bf16 hi/lo planes (columns N..Np-1 zero-filled) instead of fp32
    unsigned short* Ph; unsigned short* Pl; int64_t pitchP; int Np;
    // EPI_INTERLEAVE: GEMM row m' = m*gdiv + g addresses C[(m'/gdiv)*ldc_m + (m'%gdiv) + n*ldc_n] (gdiv = G, ldc_n = G)
    int gdiv;
    // EPI_F16F6: the result is written as f16 + block-scaled fp6 planes (cti_f16f6.h) -- the operand format of the f16f6 mode-3 product;
    // columns N..Np-1 (Np a multiple of 32) zero, logical row m -> plane row f6_prow(F6, m)
    F6Planes F6;
};
enum { EPI_F32 = 0, EPI_PLANES = 1, EPI_INTERLEAVE2 = 2, EPI_INTERLEAVE = 3, EPI_F16F6 = 4 };

__device__ __forceinline__ bf16x8 frag(const char* lds_plane, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(lds_plane + row * ROW_BYTES + ((chunk ^ ((row >> 3) & 1)) << 4));
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Tile geometry: WM x WN consumer waves, each TM x TN MFMA tiles of 32 x 32; LW loader waves; NST ring slots (16-deep
// K slices), SPB slots consumed per barrier.
template <int WM_, int WN_, int TM_, int TN_, int LW_, int NST_, int SPB_, int PL_ = 2>
struct Geo {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, LW = LW_, NST = NST_, SPB = SPB_;
    static constexpr int PL = PL_;                                     // planes per operand held in a slot: 2 (hi + lo) or 1 (plain bf16: hi only)
    static_assert(NST_ % SPB_ == 0 && NST_ > SPB_, "ring = whole groups of SPB slots, at least one group in flight");
#ifndef CTI_ALLOW_SPB4          // experiment: four slices per barrier (needs Kp % 64 == 0: NOT checked at run time -- measurement builds only)
    static_assert(32 % (16 * SPB_) == 0, "a barrier group (SPB slices of 16) must divide the planes' K padding (KPAD = 32): the K loop runs Kp / (16 SPB) groups");
#endif
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static constexpr int NCONS = WM * WN * 64, NTHR = NCONS + LW * 64;
    static constexpr int A_PLANE = BM * ROW_BYTES, B_PLANE = BN * ROW_BYTES;
    static constexpr int STAGE = PL * (A_PLANE + B_PLANE);             // [A_hi | A_lo | B_hi | B_lo]  (PL = 1: [A_hi | B_hi])
    // the staged epilogues park (TM*32) x 64 floats per consumer wave in the idle ring: a hi-only ring (PL = 1) of few slots is SMALLER than that -- the launch
    // must ask for the larger of the two (out-of-range LDS writes are dropped and reads return 0: with 4 or 6 slots of 16 KiB the last waves' rows of every
    // 256 x 256 tile came out as bias only in the plain-bf16 mode; found in round 3 by checking every row of a 9 216-row product)
    static constexpr int EPI_STAGE = WM * WN * TM * 32 * 64 * 4;
    static constexpr int LDS = NST * STAGE > EPI_STAGE ? NST * STAGE : EPI_STAGE;
};

// LDS-DMA of this wave's share of one operand's slice pair (hi plane rows, then lo plane rows: 2*ROWS/32 pieces of 1 KiB =
// 32 rows x 32 B, each contiguous in the chunk-major plane): pieces first, first + stride, ...  LDS position p (16-B chunk
// index inside the plane slice) = (row = p >> 1, c' = p & 1) receives source chunk c = c' ^ ((row >> 3) & 1) of that row.
template <int ROWS, int NPIECES, bool BOTH>
__device__ __forceinline__ void dma_pieces(const unsigned short* __restrict__ ghi, const unsigned short* __restrict__ glo,
                                           char* lds_hi, int first, int stride, int lane) {
    constexpr int PP = ROWS / 32;                                         // pieces per plane slice
#pragma unroll
    for (int u = 0; u < NPIECES; ++u) {
        const int pid = first + u * stride;                               // wave-uniform
        const bool lo = BOTH && pid >= PP;
        const int pc = (pid - (lo ? PP : 0)) * 64 + lane;
        const int row = pc >> 1, c = (pc & 1) ^ ((row >> 3) & 1);
        const unsigned short* src = (lo ? glo : ghi) + row * 16 + c * 8;
        char* dst = lds_hi + (lo ? ROWS * ROW_BYTES : 0) + (pc - lane) * 16;   // wave-uniform base; the hardware adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

// LDS-DMA of fp32 A rows (AF32 mode): one piece = 16 rows x 64 B (16 floats) of the slice; LDS position p (16-B chunk index) =
// (row = p >> 2, c' = p & 3) receives source chunk c = c' ^ ((row >> 2) & 3).  Chunks at or beyond K read the row start instead
// (any finite data: the matching K-tail of the B planes is zero).  Unlike planes the caller's matrix has no slack rows.
template <int NPIECES>
__device__ __forceinline__ void dma_pieces_f32(const float* __restrict__ arow0, int64_t ld, int k0, int K, int rows_valid, char* lds,
                                               int first, int stride, int lane) {
#pragma unroll
    for (int u = 0; u < NPIECES; ++u) {
        const int pc = (first + u * stride) * 64 + lane;
        const int row = pc >> 2, c = (pc & 3) ^ ((row >> 2) & 3);
        const int k = k0 + c * 4;
        const int rr = row < rows_valid ? row : rows_valid - 1;        // tile rows past the matrix re-read its last row (outputs discarded)
        const float* src = arow0 + (int64_t)rr * ld + (k < K ? k : 0);
        char* dst = lds + (pc - lane) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

template <int TERMS, int EPI, class G, bool AF32 = false>
__global__ __launch_bounds__(G::NTHR) void gemm_planes_kernel(PlaneGemmP p) {
    constexpr int WM = G::WM, WN = G::WN, TM = G::TM, TN = G::TN, LW = G::LW, NST = G::NST, SPB = G::SPB;
    constexpr int BM = G::BM, BN = G::BN, A_PLANE = G::A_PLANE, B_PLANE = G::B_PLANE, SLOT = G::STAGE;
    constexpr int NG = NST / SPB;                                       // ring length in barrier groups
    constexpr int NW = LW > 0 ? LW : WM * WN;                           // waves that issue the DMA
    constexpr int NPLANE = TERMS == 3 ? 2 : 1;
    static_assert(G::PL >= NPLANE, "a hi-only slot cannot feed the 3-term products");
    constexpr int PAT = (AF32 ? 2 : NPLANE) * BM / 32, PBT = NPLANE * BN / 32;       // pieces per slot of the A / B operand (hi [+ lo]; fp32 A rows: 64 B per row = 16 rows per piece)
    constexpr bool EXACT = (PAT % NW == 0) && (PBT % NW == 0);          // every issuing wave issues the same count
    constexpr int PA = (PAT + NW - 1) / NW, PB = (PBT + NW - 1) / NW;
    constexpr int NPLG = SPB * (PA + PB);                               // DMA instructions per issuing wave per barrier group
    static_assert(EXACT || TERMS == 1, "the fp32-grade configurations must split their pieces evenly over the issuing waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];        // NST slots; reused by the staged epilogue
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const bool loader = LW > 0 && wid >= WM * WN;
    const int wm = wid / WN, wn = wid % WN;
    const int tiles_n = (((EPI == EPI_PLANES || EPI == EPI_F16F6) ? p.Np : p.N) + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // Persistent tile loop: the grid is at most one workgroup per CU (a 128-KiB ring leaves room for one anyway); workgroup w
    // walks tiles w, w + grid, ... -- with the grid a multiple of 8 these keep w's XCD, so tile_coords' L2 chunking holds.
    // It removes the relaunch gap between consecutive tiles of a CU (52 tiles per CU on the mode-3 GEMM).
    for (int vtile = blockIdx.x; vtile < p.total_tiles; vtile += gridDim.x) {
    int z, tm, tn;
    tile_coords(vtile, p.total_tiles, tiles_m, tiles_n, z, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int b1 = z / p.nb2, b2 = z % p.nb2;
    const int64_t pitchA = p.pitchA, pitchB = p.pitchB;                 // locals: lambdas must not capture the argument struct
    const unsigned short* Ah = p.Ah + (b1 * p.rA1 + b2 * p.rA2 + m0) * 16 + b2 * p.kc2 * pitchA;
    const unsigned short* Al = p.Al + (b1 * p.rA1 + b2 * p.rA2 + m0) * 16 + b2 * p.kc2 * pitchA;
    const float* Af = AF32 ? p.Af + (b1 * p.rA1 + b2 * p.rA2 + m0) * p.ldaf + b2 * p.kc2 * 16 : nullptr;      // split-K: batch b2 starts kc2 * b2 chunks of 16 into the rows
    const int64_t ldaf = p.ldaf; const int Kreal = p.Kreal - (AF32 ? (int)(b2 * p.kc2 * 16) : 0), rows_valid = p.M - m0;
    const unsigned short* Bh = p.Bh + (b1 * p.rB1 + b2 * p.rB2 + n0) * 16 + b2 * p.kc2 * pitchB;
    const unsigned short* Bl = p.Bl + (b1 * p.rB1 + b2 * p.rB2 + n0) * 16 + b2 * p.kc2 * pitchB;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int iw = LW > 0 ? wid - WM * WN : wid;                        // index among the issuing waves
    // one barrier group = SPB consecutive 16-deep K slices; slot layout [A_hi | A_lo | B_hi | B_lo]
    auto issue_group = [=](int ring_pos, int grp) {
#pragma unroll
        for (int s2 = 0; s2 < SPB; ++s2) {
            char* s = smem + (ring_pos * SPB + s2) * SLOT;
            const int64_t kc = (int64_t)grp * SPB + s2;
            if (EXACT) {
                if (AF32) dma_pieces_f32<PA>(Af, ldaf, (int)kc * 16, Kreal, rows_valid, s, iw, NW, lane);   // pieces of 16 rows x 64 B
                else      dma_pieces<BM, PA, TERMS == 3>(Ah + kc * pitchA, Al + kc * pitchA, s, iw, NW, lane);
                dma_pieces<BN, PB, TERMS == 3>(Bh + kc * pitchB, Bl + kc * pitchB, s + G::PL * A_PLANE, iw, NW, lane);
            } else {                                                     // uneven split (plain-bf16 mode only): waves beyond the piece count idle
#pragma unroll
                for (int u = 0; u < PA; ++u) if (iw + u * NW < PAT) {
                    if (AF32) dma_pieces_f32<1>(Af, ldaf, (int)kc * 16, Kreal, rows_valid, s, iw + u * NW, NW, lane);
                    else      dma_pieces<BM, 1, TERMS == 3>(Ah + kc * pitchA, Al + kc * pitchA, s, iw + u * NW, NW, lane);
                }
#pragma unroll
                for (int u = 0; u < PB; ++u) if (iw + u * NW < PBT) dma_pieces<BN, 1, TERMS == 3>(Bh + kc * pitchB, Bl + kc * pitchB, s + G::PL * A_PLANE, iw + u * NW, NW, lane);
            }
        }
    };

    // Ring protocol.  Step g: the issuing waves retire group g with a COUNTED vmcnt (groups g+1 .. g+NG-2 stay in flight),
    // the raw barrier makes it visible to every wave and proves that ring position (g-1) % NG, read one step ago, is idle;
    // then group g+NG-1 is issued into that position and the consumers run their MFMAs.  No vmcnt(0) inside the loop
    // (except for the last groups), no __syncthreads() (it would drain the DMA queue).
    const int ngr = p.Kp / (BK * SPB);
    const bool issuer = LW == 0 || loader;
    if (issuer) {
#pragma unroll
        for (int i = 0; i < NG - 1; ++i) if (i < ngr) issue_group(i, i);
    }
    const int r = lane & 31, h = lane >> 5;
    // PIPE (SPB == 1, NG >= 4): fragments of slice g+1 are read from LDS at the END of step g (behind slice g's MFMAs), so
    // every step opens with MFMAs instead of an exposed LDS round trip.  For that slot g+1 must already be visible after
    // barrier g: the issuing waves retire one group more per step (NG-3 instead of NG-2 groups stay in flight).
    constexpr bool PIPE2 = (CTI_PIPE == 2) && TERMS == 1 && !AF32 && SPB == 1 && NG >= 4 && LW == 0 && TM * TN >= TM + TN;
    constexpr bool PIPE = ((CTI_PIPE == 1) || PIPE2) && SPB == 1 && NG >= 4;
    bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
    auto load_frags = [&](const char* s) {
        const char* sAh = s + (wm * TM * 32) * ROW_BYTES;
        const char* sAl = s + A_PLANE + (wm * TM * 32) * ROW_BYTES;
        const char* sBh = s + G::PL * A_PLANE + (wn * TN * 32) * ROW_BYTES;
        const char* sBl = s + G::PL * A_PLANE + B_PLANE + (wn * TN * 32) * ROW_BYTES;
        if (AF32) {
            // the slot's A region holds fp32 [row][16]: read this lane's 8 floats (two swizzled 16-B chunks) and split them
            // here -- ~24 VALU per fragment pair, issued in the slots the MFMAs leave free
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * TM * 32 + i * 32 + r;
                const char* rp = s + row * 64;
                // plain vector type, NOT HIP's float4 struct: a struct-typed LDS read may alias the LDS-DMA stores as far as hipcc can tell, and it
                // then drains vmcnt(0) -- every DMA in flight -- in front of the read (found with the f16f6 kernel; this is what made this path "neutral")
                const f6_f32x4 x0 = *reinterpret_cast<const f6_f32x4*>(rp + (((2 * h) ^ ((row >> 2) & 3)) << 4));
                const f6_f32x4 x1 = *reinterpret_cast<const f6_f32x4*>(rp + (((2 * h + 1) ^ ((row >> 2) & 3)) << 4));
                const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const __bf16 hv = static_cast<__bf16>(xs[e]);
                    ah[i][e] = hv;
                    if (TERMS == 3) al[i][e] = static_cast<__bf16>(xs[e] - static_cast<float>(hv));
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) { ah[i] = frag(sAh, i * 32 + r, h); if (TERMS == 3) al[i] = frag(sAl, i * 32 + r, h); }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bh[j] = frag(sBh, j * 32 + r, h); if (TERMS == 3) bl[j] = frag(sBl, j * 32 + r, h); }
    };
    auto mfma_frags = [&]() {
#if CTI_ABL & 2
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
#else
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (TERMS == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
#endif
    };
#ifndef CTI_X3_STAGGER
#define CTI_X3_STAGGER 0
#endif
    // STAGGER (one slot per barrier, no loader waves; OFF here): waves w and w + NW/2 share a SIMD and run the same program; the upper half
    // reads slice g BEFORE barrier g + 1 and issues its MFMAs AFTER it, so that one SIMD partner computes while the other reads.  It pays in
    // cti_gemm_f16f6.hip (1.30 -> 1.14 ms on that kernel's MFMA + LDS part: 18 MFMAs behind 24 LDS reads per step); here (24 MFMAs behind 12
    // reads of 16-deep slices) the whole forward measured 5.48 ms with it against 5.31 ms without, same box, interleaved rounds.
    constexpr bool STAG = (CTI_X3_STAGGER != 0) && LW == 0 && SPB == 1 && !PIPE;
    const bool lag = STAG && __builtin_amdgcn_readfirstlane(wid) >= (WM * WN) / 2;
    auto sync_g = [&](int g) {
        if (issuer) {
            const int rem = ngr - 1 - g;                                 // groups issued after group g so far: min(NG-2, rem)
            if (!EXACT) wait_vmcnt<0>();
            else if (PIPE) { if (rem >= NG - 2) wait_vmcnt<(NG - 3) * NPLG>(); else wait_vmcnt<0>(); }   // groups <= g+1 landed
            else if (NG >= 3 && rem >= NG - 2) wait_vmcnt<(NG - 2) * NPLG>();
            else if (NG >= 4 && rem == 1) wait_vmcnt<NPLG>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
    };
    if (lag) {
        int pos = 0;
        sync_g(0);
        if (issuer && NG - 1 < ngr && !(CTI_ABL & 1)) issue_group(NG - 1, NG - 1);
        for (int g = 0; g < ngr; ++g) {
            load_frags(smem + pos * SLOT);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pos = pos == NG - 1 ? 0 : pos + 1;
            if (g + 1 < ngr) {
                sync_g(g + 1);
                if (issuer && g + NG < ngr && !(CTI_ABL & 1)) issue_group(pos == 0 ? NG - 1 : pos - 1, g + NG);
            } else {
                __syncthreads();                  // pairs with the lead waves' closing barrier
            }
            mfma_frags();
        }
    } else if (PIPE2) {
        // Round-3 experiment (CTI_PIPE=2, plain bf16): TWO fragment sets, the LDS reads of slice g + 1 interleaved one-for-one with the MFMAs of slice g by
        // sched_group_barrier -- the software-pipelined stream the vendor GEMMs are written in, as far as the compiler can be told to emit it.
        bf16x8 fa[2][TM], fb[2][TN];
        auto ldf = [&](const char* s, bf16x8 (&a)[TM], bf16x8 (&b)[TN]) {
            const char* sAh = s + (wm * TM * 32) * ROW_BYTES;
            const char* sBh = s + G::PL * A_PLANE + (wn * TN * 32) * ROW_BYTES;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = frag(sAh, i * 32 + r, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = frag(sBh, j * 32 + r, h);
        };
        auto body = [&](int g, bf16x8 (&ca)[TM], bf16x8 (&cb)[TN], bf16x8 (&na)[TM], bf16x8 (&nb)[TN]) {
            const int pos = g % NG;
            if (g > 0) {
                sync_g(g);
                if (g + NG - 1 < ngr) issue_group(pos == 0 ? NG - 1 : pos - 1, g + NG - 1);
            }
            if (g + 1 < ngr) ldf(smem + (pos == NG - 1 ? 0 : pos + 1) * SLOT, na, nb);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < TM + TN; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
        };
        sync_g(0);
        if (NG - 1 < ngr) issue_group(NG - 1, NG - 1);
        ldf(smem, fa[0], fb[0]);
        for (int g = 0; g < ngr; g += 2) {
            body(g, fa[0], fb[0], fa[1], fb[1]);
            if (g + 1 < ngr) body(g + 1, fa[1], fb[1], fa[0], fb[0]);
        }
        __syncthreads();
    } else {
    int pos = 0;
    for (int g = 0; g < ngr; ++g) {
        sync_g(g);
        if (issuer && g + NG - 1 < ngr && !(CTI_ABL & 1)) issue_group(pos == 0 ? NG - 1 : pos - 1, g + NG - 1);
        if (!loader) {
            if (PIPE) {
                if (g == 0) load_frags(smem + pos * SLOT);
                mfma_frags();
                if (g + 1 < ngr) load_frags(smem + (pos == NG - 1 ? 0 : pos + 1) * SLOT);
            } else {
#pragma unroll
                for (int s2 = 0; s2 < SPB; ++s2) { load_frags(smem + (pos * SPB + s2) * SLOT); mfma_frags(); }
            }
        }
        pos = pos == NG - 1 ? 0 : pos + 1;
    }
    __syncthreads();                              // every wave is done reading the ring before the epilogue reuses it
    }
    if (!loader) {

    const int64_t boff = b1 * p.sC1 + b2 * p.sC2;
    do {
    if (EPI == EPI_F16F6) {
        // Per 32-column MFMA tile the wave parks its (TM*32) x 32 block in a private LDS patch ([row][36 floats]: conflict-free 16-B row reads),
        // scale / bias / ReLU applied on the way in; then every lane encodes ONE (row, 32-block) item straight into the f16f6 planes.
        static_assert(EPI != EPI_F16F6 || TM * 32 == 64, "one (row, block) item per lane");
        float* stg = reinterpret_cast<float*>(smem) + wid * (64 * 36);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + (wn * TN + j) * 32, n = nb + (lane & 31);
            const bool real = n < p.N;
            const float sc = (real && p.scale) ? p.scale[b1 * p.scale_bs + n / p.scale_div] : 1.f;
            const float bi = (real && p.bias) ? p.bias[b1 * p.bias_bs + n] : 0.f;
            if (j) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous tile's reads are done before overwriting
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    float x = acc[i][j][e] * sc + bi;
                    if (p.relu) x = relu_nan(x);
                    stg[row * 36 + (lane & 31)] = real ? x : 0.f;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // same wave writes and reads its patch: in-order LDS, no barrier
            const int m = m0 + wm * TM * 32 + lane;
            if (m < p.M && nb < p.Np) f6_encode_row32_lds(stg + lane * 36, p.F6, f6_prow(p.F6, boff + m), nb >> 5);
        }
        break;
    }
    if (EPI == EPI_PLANES || EPI == EPI_F32) {
        // Staged epilogue, 64 columns of the wave's sub-tile at a time.  The wave parks a (TM*32) x 64 fp32 block in its own
        // slice of the (now idle) LDS ([row][16 slots of 16 B], slot ^= row & 1: conflict-free for the ds_write_b32 column
        // writes and the ds_read_b128 row reads); then every lane owns 8 consecutive columns of a row: scale/bias/ReLU and
        // 16-B stores.  EPI_PLANES: a lane's 8 columns are half a 16-wide chunk: 16 B per row and chunk, rows 32 B apart.
        float* stg = reinterpret_cast<float*>(smem) + wid * (TM * 32 * 64);
        const int c8 = lane & 7;
        const bool vecC = (EPI == EPI_F32) && (p.ldc_n == 1) && ((p.ldc_m & 3) == 0) && ((boff & 3) == 0) &&
                          ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
        const int ncols = (EPI == EPI_PLANES) ? p.Np : p.N;
#pragma unroll
        for (int jp = 0; jp < TN; jp += 2) {
            if (jp) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // previous pass's reads are done before overwriting
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        const int col = jj * 32 + (lane & 31);
                        stg[row * 64 + ((((col >> 2) ^ (row & 1))) << 2) + (col & 3)] = acc[i][jp + jj][e];
                    }
            // each wave reads back only its own slice: LDS operations of one wave complete in order, no barrier needed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int nb = n0 + wn * TN * 32 + jp * 32 + c8 * 8;            // first of this lane's 8 columns
            float sc[8], bi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = nb + u;
                const bool real = n < p.N;
                sc[u] = (real && p.scale) ? p.scale[b1 * p.scale_bs + n / p.scale_div] : 1.f;
                bi[u] = (real && p.bias) ? p.bias[b1 * p.bias_bs + n] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < TM * 4; ++it) {
                const int row = it * 8 + (lane >> 3);
                const int m = m0 + wm * TM * 32 + row;
                const float4 x0 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * c8) ^ (row & 1)) << 2));
                const float4 x1 = *reinterpret_cast<const float4*>(stg + row * 64 + (((2 * c8 + 1) ^ (row & 1)) << 2));
                if (m >= p.M || nb >= ncols) continue;
                float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x[u] = x[u] * sc[u] + bi[u];
                    if (p.relu) x[u] = relu_nan(x[u]);
                    if (nb + u >= p.N) x[u] = 0.f;
                }
                if (EPI == EPI_PLANES) {
                    unsigned short hb[8], lb[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { hb[u] = bf16_bits(x[u]); lb[u] = bf16_bits(x[u] - bf16_to_f32(hb[u])); }
                    const int64_t o = (int64_t)(nb >> 4) * p.pitchP + (boff + m) * 16 + (nb & 15);
                    *reinterpret_cast<uint4*>(p.Ph + o) = pack8(hb);
                    *reinterpret_cast<uint4*>(p.Pl + o) = pack8(lb);
                } else {
                    float* dst = p.C + boff + (int64_t)m * p.ldc_m + (int64_t)nb * p.ldc_n;
                    if (vecC && nb + 8 <= p.N) {
                        reinterpret_cast<float4*>(dst)[0] = make_float4(x[0], x[1], x[2], x[3]);
                        reinterpret_cast<float4*>(dst)[1] = make_float4(x[4], x[5], x[6], x[7]);
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (nb + u < p.N) dst[(int64_t)u * p.ldc_n] = x[u];
                    }
                }
            }
        }
        break;
    }
    float* C = p.C + boff;
    if (EPI == EPI_INTERLEAVE2) {
        // GEMM rows (2m, 2m+1) are the two glimpses of one (v,q) row: registers e and e+1 (e even) of a lane are
        // adjacent floats in out[b, vq, a, 0:2] -> one 8-B store per lane, 256 contiguous bytes per half-wave
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            if (n >= p.N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);   // even
                    if (m < p.M && !((CTI_ABL & 4) && acc[i][j][e] != 12345.f)) {
                        float2 v2 = make_float2(acc[i][j][e], acc[i][j][e + 1]);
                        *reinterpret_cast<float2*>(C + (int64_t)(m >> 1) * p.ldc_m + (int64_t)n * 2) = v2;
                    }
                }
            }
        }
        break;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (n >= p.N) continue;
        const float sc = p.scale ? p.scale[b1 * p.scale_bs + n / p.scale_div] : 1.f;
        const float bi = p.bias ? p.bias[b1 * p.bias_bs + n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m < p.M) {
                    float x = acc[i][j][e] * sc + bi;
                    if (p.relu) x = relu_nan(x);
                    C[(int64_t)(m / p.gdiv) * p.ldc_m + (m % p.gdiv) + (int64_t)n * p.ldc_n] = x;
                }
            }
        }
    }
    } while (0);
    }                                             // !loader
    __syncthreads();                              // the staged epilogue's LDS slices are free again before the next tile's DMA
    }                                             // persistent tile loop
}

// split-K reduce + epilogue: C[m, n] = act(scale[n / div] * sum_s part[s][m][n] + bias[n]); one float4 of a row per thread
// (round 4: nb batches -- rows z * M + m of the partials, C / scale / bias advanced by sC1 / scale_bs / bias_bs per batch)
__global__ __launch_bounds__(256) void ksplit_reduce_kernel(const float* __restrict__ part, int S, int M, int N, float* __restrict__ C,
                                                            int64_t ldc_m, const float* __restrict__ scale, int scale_div,
                                                            const float* __restrict__ bias, int relu, int nb, int64_t sC1, int64_t scale_bs, int64_t bias_bs) {
    const int n4 = (N + 3) >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)nb * M * n4) return;
    const int mz = (int)(idx / n4), n0 = (int)(idx % n4) * 4;
    const int z = mz / M, m = mz - z * M;
    const int64_t MN = (int64_t)nb * M * N;
    const float* src = part + (int64_t)mz * N + n0;
    C += (int64_t)z * sC1;
    if (scale) scale += (int64_t)z * scale_bs;
    if (bias) bias += (int64_t)z * bias_bs;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (N & 3) == 0;
    for (int s = 0; s < S; ++s) {
        if (vec) {
            const float4 x = *reinterpret_cast<const float4*>(src + s * MN);
            a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (n0 + u < N) a[u] += src[s * MN + u];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = n0 + u;
        if (n < N) {
            float x = a[u] * (scale ? scale[n / scale_div] : 1.f) + (bias ? bias[n] : 0.f);
            if (relu) x = relu_nan(x);
            C[(int64_t)m * ldc_m + n] = x;
        }
    }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

#ifndef CTI_LW
#define CTI_LW 0
#endif
// tile configurations (ring slots are 16-deep K slices):
//   big   256 x 256, 4 slots of 32 KiB, 1 per barrier (3 in flight);  mid 256 x 128, 6 slots of 24 KiB, 2 per barrier;
//   small 128 x 128, 6 slots of 16 KiB, 2 per barrier
#ifndef CTI_BIG_SPB
#define CTI_BIG_SPB 1
#endif
using GeoBig = Geo<4, 2, 2, 4, CTI_LW, 4, CTI_BIG_SPB>;
// plain-bf16 products (one MFMA per product): slots hold the hi planes only (16 KiB per 16-deep slice of the 256 x 256 tile).  Measured at the
// configs[1] mode-3 shape (tools/tune_gemm.py run 5 2): 4 slots / 1 per barrier 1.52 ms, 6 / 2 1.54 ms, deeper rings slower -- the kernel is
// not ring-latency-bound; its DMA (0.5 ms), MFMA (0.7 ms) and store (0.4 ms) phases add up instead of overlapping, as in the 3-term form.
// Round 3, at the shapes this geometry actually runs in the models (the hoisted v projections of configs[2] / [3]: 9 216 x 2 048 against n x 1 024 rows,
// tools/bench_gemm_pb.py, split of x included): 4 / 1: 205.6 / 202.5 / 615.9 us (n = 3 as one 3 072-row weight, n = 3, n = 11); 8 / 2: 200.9 / 195.8 / 597.1;
// 6 / 2: 201.6 / 197.3 / 594.5 (715 TFLOP/s = 0.29 of the bf16 peak); 8 / 1: 209.9 / 203.9 / 620.5; 6 / 1 with the fragment reads of the next slice behind the
// MFMAs (CTI_PIPE): 227.5 / 221.3 / 678.9.  Two slices per barrier it is.  (9 / 3 "measured" 582.8 -- by dropping the K tail: planes are padded to 32 in K,
// a barrier group must divide that; the static_assert in Geo now says so.)
#ifndef CTI_BIG1_NST
#define CTI_BIG1_NST 6
#endif
#ifndef CTI_BIG1_SPB
#define CTI_BIG1_SPB 2
#endif
#ifndef CTI_BIG1_WM            // wave grid and wave tile of the plain-bf16 256 x 256 geometry (experiment knobs: 2,2,4,4 = four waves of 128 x 128, one per SIMD)
#define CTI_BIG1_WM 4
#define CTI_BIG1_WN 2
#define CTI_BIG1_TM 2
#define CTI_BIG1_TN 4
#endif
using GeoBig1 = Geo<CTI_BIG1_WM, CTI_BIG1_WN, CTI_BIG1_TM, CTI_BIG1_TN, CTI_LW, CTI_BIG1_NST, CTI_BIG1_SPB, 1>;
using GeoMid = Geo<4, 2, 2, 2, CTI_LW, 6, 2>;
using GeoSmall = Geo<2, 2, 2, 2, (CTI_LW > 2 ? 2 : CTI_LW), 6, 2>;

template <int TERMS, int EPI, class G, bool AF32 = false>
int launch_cfg(const PlaneGemmP& p, long long nb, int ncols, hipStream_t st) {
    auto kern = gemm_planes_kernel<TERMS, EPI, G, AF32>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail((int)e, "gemm_nt_planes: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    const long long total = nb * ((p.M + G::BM - 1) / G::BM) * ((ncols + G::BN - 1) / G::BN);
    if (total > 0x7fffffffLL) return fail(CTI_E_SHAPE, "gemm_nt_planes: %lld tiles exceed the grid", total);
    PlaneGemmP q = p;
    q.total_tiles = (int)total;
    static thread_local int n_cu = 0;
    if (n_cu == 0) { (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
#ifndef CTI_PERSIST
#define CTI_PERSIST 1
#endif
    const int per_cu = G::LDS > 80 * 1024 ? 1 : 2;                      // workgroups that fit a CU's 160 KiB of LDS
    long long grid = CTI_PERSIST ? (long long)n_cu * per_cu : total;
    if (grid > total) grid = total;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NTHR), G::LDS, st, q);
    return launch_status("gemm_nt_planes");
}

template <int TERMS, int EPI>
int launch_epi(const PlaneGemmP& p, long long nb, int ncols, int cfg, hipStream_t st) {
    if (p.Af) {                                              // fp32 A operand: the row-major epilogues; the slot's A region holds 64 B per row (= both planes' room)
        if constexpr (TERMS == 3 && (EPI == 0 || EPI == 1 || EPI == 4)) {
            switch (cfg) {
                case 2: return launch_cfg<TERMS, EPI, GeoBig, true>(p, nb, ncols, st);
                case 1: return launch_cfg<TERMS, EPI, GeoMid, true>(p, nb, ncols, st);
                default: return launch_cfg<TERMS, EPI, GeoSmall, true>(p, nb, ncols, st);
            }
        } else if constexpr (TERMS == 1 && EPI == 0) {
            // plain bf16 (round 4): the batch-sized GEMMs of the model forwards read their fp32 activations directly -- no split launch in front of
            // every product (41 per FFOE forward).  The hi-only 256 x 256 geometry has no room for fp32 rows: the 256 x 128 tile takes its place.
            if (cfg >= 1) return launch_cfg<TERMS, EPI, GeoMid, true>(p, nb, ncols, st);
            return launch_cfg<TERMS, EPI, GeoSmall, true>(p, nb, ncols, st);
        } else {
            return fail(CTI_E_UNSUPPORTED, "gemm_nt_planes: fp32 A operand with terms=%d epi=%d", TERMS, EPI);
        }
    }
    switch (cfg) {
        case 2: if constexpr (TERMS == 1) return launch_cfg<TERMS, EPI, GeoBig1>(p, nb, ncols, st); else return launch_cfg<TERMS, EPI, GeoBig>(p, nb, ncols, st);
        case 1: return launch_cfg<TERMS, EPI, GeoMid>(p, nb, ncols, st);
        default: return launch_cfg<TERMS, EPI, GeoSmall>(p, nb, ncols, st);
    }
}

}  // namespace

int planes_kp(int K) { return round_up(K, KPAD); }

// Skinny GEMMs (the GRU's recurrent products, the classifier, the residual projections: M = batch size) give the 256 CUs only a
// few dozen 128 x 128 tiles, each walking the whole K: split K so that about one workgroup per CU runs, each over >= 128 of K.
int plan_ksplit(int M, int N, int Kp, long long nb) {
#ifdef CTI_NO_KSPLIT
    return 1;
#endif
    if (nb < 1 || Kp < 256) return 1;
    const long long tiles = nb * (long long)((M + 127) / 128) * ((N + 127) / 128);       // (round 4: batches of skinny products split too)
    if (tiles > 96) return 1;
    int best = 1;
    for (int s = 2; s <= 16; ++s) {
        if (Kp % (s * 32) != 0 || Kp / s < 128) continue;
        if (tiles * s > 288) break;
        best = s;
    }
    return best;
}
size_t planes_bytes(int64_t rows_alloc, int K) { return 2 * sizeof(unsigned short) * (size_t)rows_alloc * planes_kp(K); }

int split_planes(const float* x, int64_t ld, int64_t rows, int K, unsigned short* hi, unsigned short* lo, int64_t rows_alloc,
                 hipStream_t st) {
    const int Kp = planes_kp(K);
    const int64_t n = rows * 4;
#if CTI_SPLIT_KFAST
    const int64_t blocks = (n + 255) / 256;
    const unsigned gz = (unsigned)((blocks + 65534) / 65535);              // row blocks spread over (y, z): y <= 65535
    const unsigned gy = (unsigned)((blocks + gz - 1) / gz);
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)(Kp >> 5), gy, gz), dim3(256), 0, st, x, ld, rows, K, Kp, hi, lo, rows_alloc * 16);
#else
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(Kp >> 5)), dim3(256), 0, st, x, ld, rows, K, Kp, hi, lo,
                       rows_alloc * 16);
#endif
    return launch_status("split_planes");
}

int split_planes16(const unsigned short* x, int64_t ld, int64_t rows, int K, unsigned short* hi, unsigned short* lo, int64_t rows_alloc, hipStream_t st) {
    if (rows <= 0 || K <= 0 || (K & 7) || (ld & 7) || (reinterpret_cast<uintptr_t>(x) & 15)) return fail(CTI_E_ALIGN, "split_planes16: rows=%lld K=%d ld=%lld (K, ld multiples of 8; 16-B aligned)", (long long)rows, K, (long long)ld);
    const int Kp = planes_kp(K);
    const int64_t blocks = (rows * 4 + 255) / 256;
    const unsigned gz = (unsigned)((blocks + 65534) / 65535);
    const unsigned gy = (unsigned)((blocks + gz - 1) / gz);
    hipLaunchKernelGGL(split16_kernel, dim3((unsigned)(Kp >> 5), gy, gz), dim3(256), 0, st, x, ld, rows, K, hi, lo, rows_alloc * 16);
    return launch_status("split_planes16");
}

int split_planes_t(const float* x, int64_t ld, int64_t M, int n, int64_t Mp, unsigned short* hi, unsigned short* lo, int64_t rows_alloc,
                   hipStream_t st) {
    if (Mp % 16 != 0 || Mp / 16 > 65535) return fail(CTI_E_SHAPE, "split_planes_t: depth %lld", (long long)Mp);
    hipLaunchKernelGGL(split_t_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(Mp / 16)), dim3(256), 0, st, x, ld, M, n, hi, lo, rows_alloc * 16);
    return launch_status("split_planes_t");
}

// K ranges for C = a^T b (contraction over the M rows): about two workgroups per CU, at least 256 of depth each
int plan_ksplit_tn(int64_t M, int N, int K) {
    // CTI_TN_PLAN=0: the first planner (128 x 128 tiles in mind).  Default: for outputs of at least 64 tiles of 256 x 256 over a deep
    // contraction (measured, tools/bench_gemm_tn.py: 9216 rows, 3072 x 2048 out: 829 -> 462 us; at 32 tiles and below the first planner is
    // 3-10 % ahead), plan for the 256 x 256 tile (twice the operand reuse per LDS byte; gemm_nt_planes picks it once tiles x S >= 256): choose the S that
    // minimises  rounds(S) * depth(S) + the partials' write + read,  rounds = ceil(tiles * S / 256 workgroups).
    static const int mode = [] { const char* e = getenv("CTI_TN_PLAN"); return e ? atoi(e) : 1; }();
    const long long t256 = (long long)((N + 255) / 256) * ((K + 255) / 256);
    if (mode != 0 && t256 >= 64 && M >= 2048) {
        const long long smax = M / 512 < 32 ? M / 512 : 32;
        const double pen = (double)N * K * 2e-5;                  // one partial's write + read, in units of one depth step of a 256 x 256 tile
        double best_cost = 1e300; int best = 1;
        for (long long sx = 1; sx <= smax; ++sx) {
            const long long rounds = (t256 * sx + 255) / 256;
            const double cost = (double)rounds * ((double)M / sx) + (sx > 1 ? pen * sx : 0.0);
            if (cost < best_cost) { best_cost = cost; best = (int)sx; }
        }
        if (t256 * best >= 256) return best;                      // otherwise the 256-tile would not be chosen anyway: fall through
    }
    const long long tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
    long long s = 512 / (tiles > 0 ? tiles : 1);
    const long long smax = (M + 255) / 256;
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

int gemm_nt_planes(const PlaneGemmArgs& a, hipStream_t st) {
    PlaneGemmP p{};
    p.Ah = a.Ah; p.Al = a.Al; p.Bh = a.Bh; p.Bl = a.Bl; p.C = a.C;
    p.pitchA = a.rows_allocA * 16; p.pitchB = a.rows_allocB * 16; p.ldc_m = a.ldc_m; p.ldc_n = a.ldc_n;
    p.rA1 = a.rA1; p.rA2 = a.rA2; p.rB1 = a.rB1; p.rB2 = a.rB2; p.sC1 = a.sC1; p.sC2 = a.sC2;
    p.nb2 = a.nb2; p.M = a.M; p.N = a.N; p.Kp = a.Kp;
    p.scale = a.scale; p.scale_div = a.scale_div > 0 ? a.scale_div : 1; p.bias = a.bias; p.relu = a.relu;
    p.scale_bs = a.scale_bs; p.bias_bs = a.bias_bs;
    p.Ph = a.Ph; p.Pl = a.Pl; p.pitchP = a.rows_allocP * 16; p.Np = a.Np; p.gdiv = a.gdiv > 0 ? a.gdiv : 1;
    p.Af = a.Af; p.ldaf = a.ldaf; p.Kreal = a.Kreal;
    if (a.f6out) p.F6 = *a.f6out;
    // CTI_GEMM_TRACE=1 (debugging aid): one line per product on stderr -- which shapes a model forward really runs
    static const bool trace = [] { const char* e = getenv("CTI_GEMM_TRACE"); return e && e[0] == '1'; }();
    if (trace) fprintf(stderr, "gemm_nt_planes M=%d N=%d Kp=%d nb=%dx%d terms=%d epi=%d ksplit=%d Af=%d Abf=%d scale=%d bias=%d relu=%d\n", a.M, a.N, a.Kp, a.nb1, a.nb2, a.terms, a.epi,
                       a.ksplit, a.Af != nullptr, a.Abf != nullptr, a.scale != nullptr, a.bias != nullptr, a.relu);
    // round 6: a product its caller planned a split-K for (few rows, long K) is ONE launch of cti_gemm_skinny.hip instead of K ranges + partials + a reduce launch
    // (CTI_GEMM_SKINNY=0: off, A/B; a forced tile geometry -- the tests' cti_set_tuning -- keeps the split-K path reachable)
    if ((a.ksplit > 1 || a.M <= 512) && tuning_gemm_cfg() < 0 && gemm_skinny_eligible(a)) return gemm_skinny(a, st);     // (any product of at most 512 rows: planned split or not)
    if (a.ksplit > 1) {
        if (a.nb1 < 1 || a.nb2 != 1 || a.epi != 0 || !a.partial || a.ldc_n != 1 || a.Kp % (a.ksplit * KPAD) != 0)
            return fail(CTI_E_UNSUPPORTED, "gemm_nt_planes: split-K needs one fp32 row-major GEMM (ksplit=%d Kp=%d epi=%d)", a.ksplit, a.Kp, a.epi);
        PlaneGemmArgs b = a;
        b.ksplit = 1; b.partial = nullptr;
        b.nb2 = a.ksplit; b.rA2 = 0; b.rB2 = 0; b.Kp = a.Kp / a.ksplit;
        b.C = a.partial; b.ldc_m = a.N; b.ldc_n = 1; b.sC1 = (int64_t)a.M * a.N; b.sC2 = (int64_t)a.nb1 * a.M * a.N;       // partials [K range][batch][M][N]
        b.scale = nullptr; b.bias = nullptr; b.relu = 0;
        b.kc2 = a.Kp / a.ksplit / 16;
        int rc = gemm_nt_planes(b, st); if (rc) return rc;
        if (a.partials_only) return CTI_OK;                   // the caller's own kernel reduces them (cti_linear_residual_pb)
        const int64_t items = (int64_t)a.nb1 * a.M * ((a.N + 3) / 4);
        hipLaunchKernelGGL(ksplit_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, a.partial, a.ksplit, a.M, a.N, a.C,
                           a.ldc_m, a.scale, a.scale_div > 0 ? a.scale_div : 1, a.bias, a.relu, a.nb1, a.sC1, a.scale_bs, a.bias_bs);
        return launch_status("gemm_nt_planes/ksplit_reduce");
    }
    p.kc2 = a.kc2;
    if (a.Af && ((a.ldaf & 3) || (a.Kreal & 3) || (reinterpret_cast<uintptr_t>(a.Af) & 15)))
        return fail(CTI_E_ALIGN, "gemm_nt_planes: fp32 A operand needs 16-B aligned rows and K %% 4 == 0 (ld=%lld K=%d)", (long long)a.ldaf, a.Kreal);
    if (a.Kp % KPAD != 0) return fail(CTI_E_ALIGN, "gemm_nt_planes: Kp=%d is not a multiple of %d", a.Kp, KPAD);
    const int ncols = (a.epi == 1 || a.epi == 4) ? a.Np : a.N;
    if (a.epi == 4 && (!a.f6out || a.Np % 32 != 0 || a.terms != 3)) return fail(CTI_E_UNSUPPORTED, "gemm_nt_planes: f16f6 output needs planes, Np %% 32 == 0 and the 3-term mode");
    const long long nb = (long long)a.nb1 * a.nb2;
    // tile choice by a makespan model: every geometry runs one workgroup per CU, so a launch takes ceil(tiles / 256) rounds of one tile
    // time; relative tile times from the measured full-grid rates (128x128 ~0.43, 256x128 ~0.7, 256x256 ~1.0 PFLOP/s issued: operand bytes
    // through the L2 -> LDS DMA path scale with (BM + BN) / (BM * BN)): 2.33 / 2.86 / 4.0.  The first rule -- "the largest tile that still
    // gives every CU a workgroup", CTI_GEMM_CFG_MODEL=0 -- took 288 tiles of 256x128 (two rounds, the second 1/8 full) over 144 of
    // 256x256 (one round): CTI model forward 2.03 -> 1.91 ms, training step 8.1 -> 7.8 ms with the model.
    auto tiles = [&](int bm, int bn) { return nb * ((a.M + bm - 1) / bm) * ((ncols + bn - 1) / bn); };
    static const int model = [] { const char* e = getenv("CTI_GEMM_CFG_MODEL"); return e ? atoi(e) : 1; }();
    int cfg = 0;
    if (model) {
        auto span = [&](int bm, int bn, double t) { return (double)((tiles(bm, bn) + 255) / 256) * t; };
        double best = span(128, 128, 2.33);
        if (a.M > 128 && span(256, 128, 2.86) < best) { best = span(256, 128, 2.86); cfg = 1; }
        if (a.M > 128 && ncols > 128 && span(256, 256, 4.0) < best) { best = span(256, 256, 4.0); cfg = 2; }
    } else {
        if (a.M > 128 && tiles(256, 128) >= 256) cfg = 1;
        if (a.M > 128 && ncols > 128 && tiles(256, 256) >= 256) cfg = 2;
    }
#ifdef CTI_FORCE_CFG
    cfg = CTI_FORCE_CFG;
#endif
    if (tuning_gemm_cfg() >= 0) cfg = tuning_gemm_cfg();                   // cti_set_tuning(CTI_TUNE_GEMM_CFG): tests reach every geometry at small shapes
    // plain-bf16 products on the 256 x 256 tile: the round-4 kernel (cti_gemm16.hip; CTI_GEMM16=0 keeps this file's kernel: A/B)
    static const bool use16 = [] { const char* e = getenv("CTI_GEMM16"); return !(e && e[0] == '0'); }();
    // round 6: the smaller tiles' plain-bf16 products as well (G16Geo<4, 2, 8> / <8, 2, 6> in cti_gemm16.hip: c3 566 -> 522 us, profiles/r06_gemm_mid_size.txt;
    // CTI_GEMM16_SMALL=0: this file's kernel for them, A/B)
    static const bool small16 = [] { const char* e = getenv("CTI_GEMM16_SMALL"); return !(e && e[0] == '0'); }();
    if (a.Abf) {                                                     // a row-major bf16 A operand exists on cti_gemm16.hip only, whatever the tile and the A/B switches
        if (!gemm16_eligible(a)) return fail(CTI_E_UNSUPPORTED, "gemm_nt_planes: a bf16-row A operand needs a product cti_gemm16.hip takes (terms=%d epi=%d Kp=%d)", a.terms, a.epi, a.Kp);
        return gemm16_planes(a, st, cfg);
    }
    if (use16 && (cfg == 2 || small16) && gemm16_eligible(a)) return gemm16_planes(a, st, cfg);
    const int epi = (a.epi == 3 && p.gdiv == 2 && a.ldc_n == 2) ? 2 : a.epi;
    const int key = (a.terms == 3 ? 4 : 0) + epi;
    switch (key) {
        case 0: return launch_epi<1, 0>(p, nb, ncols, cfg, st);
        case 1: return launch_epi<1, 1>(p, nb, ncols, cfg, st);
        case 2: return launch_epi<1, 2>(p, nb, ncols, cfg, st);
        case 3: return launch_epi<1, 3>(p, nb, ncols, cfg, st);
        case 4: return launch_epi<3, 0>(p, nb, ncols, cfg, st);
        case 5: return launch_epi<3, 1>(p, nb, ncols, cfg, st);
        case 6: return launch_epi<3, 2>(p, nb, ncols, cfg, st);
        case 7: return launch_epi<3, 3>(p, nb, ncols, cfg, st);
        case 8: return launch_epi<3, 4>(p, nb, ncols, cfg, st);
        default: return fail(CTI_E_UNSUPPORTED, "gemm_nt_planes: epi=%d terms=%d", a.epi, a.terms);
    }
}

}  // namespace cti
