// cti_gemm16.hip -- plain-bf16 NT GEMM on chunk-major hi planes, round-4 schedule ("two wave groups, one interval apart").
//
// The plain-bf16 arithmetic mode (one v_mfma per product: precision='bf16', BASELINE configs[2] / [3]) ran on gemm_planes_kernel<1> of
// cti_gemm_bf16x3.hip: 256 x 256 tile, every wave through the same phases (LDS reads -> MFMAs) behind one barrier per 16- or 32-deep K step:
// 600-740 TFLOP/s where the vendor GEMM reaches 1 075-1 140 on the same shapes (tools/ref_gemm_rate.py; reference src/fc.py:22-29 is the
// layer this serves).  The matrix pipe idles while all eight waves read, the LDS idles while all eight multiply.  Here (prototype with
// measurements: tools/mb/mb_gemm16.hip, profiles/r04_mb_gemm16.txt -- 1 149 TFLOP/s at 4096^3, 874 without the stagger):
//   * 256 x 256 tile, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator registers), taken
//     TRANSPOSED (D = B_tile A_tile^T) so that a lane holds four consecutive columns of one output row: 16-B stores without a shuffle;
//   * an NS-slot LDS ring of 32-deep K stages (A 256 rows x 64 B | B 256 x 64 B = 32 KiB; NS = 4), filled by global_load_lds_dwordx4 straight from
//     the chunk-major planes ([K/16][rows][16]: one piece = 16 rows x 64 B gathers each row's two 32-B chunk halves); the 16-B unit c of row r
//     lands at unit c ^ ((-(r >> 2)) & 3): conflict-free ds_read_b128 fragments for the 16 x 16 x 32 operand layout (lane = row l & 15, K group l >> 4);
//   * every wave alternates a LOAD interval (12 fragment reads of the stage, its 4 DMA pieces of the stage NS - 1 ahead, the previous tile's
//     epilogue when one is pending, counted vmcnt + lgkmcnt(0)) with a COMPUTE interval (32 MFMAs), one raw s_barrier between intervals, and waves
//     4-7 -- the SIMD partners of waves 0-3 -- run ONE INTERVAL BEHIND: on every SIMD one wave's MFMAs run beside the other's LDS reads and DMA issue;
//   * a workgroup's tiles are one stream of stages (persistent, XCD-aware tile order): the ring never drains at a tile boundary;
//   * bias and weight-norm scale of the tile's 64 columns per wave travel through LDS too (two global_load_lds_dword per wave and tile, double
//     buffered): an ordinary global load in the epilogue would make the compiler drain vmcnt(0) -- the whole ring -- in front of it.
// Stream-K (round 6; reference layer src/fc.py:22-29 at the hoisted projections' shapes: 9216 x 3072 = 432 tiles = 1.69 rounds of 256 workgroups):
//   with T tiles on P workgroups (8 | P), T > P and T % P != 0, the last T - (T / P - 1) P tiles (between one and two rounds' worth) are cut PER XCD: the
//   tiles of XCD x's chunk of the tile order (tile_coords: the tiles its L2 shares operands for) that are left after the whole rounds are ONE sequence of
//   K stages cut into P / 8 equal ranges, one per workgroup of that XCD (w = x + 8 s); a range is at least one tile long, so it is [the K tail of a
//   tile][whole tiles][the K head of the next tile] and every tile has at most TWO contributors, workgroups w and w + 8.  A workgroup runs its pieces as  head piece -> whole tiles (its range's, then its data-parallel ones) -> tail piece:
//     * the head piece's accumulators go to the workgroup's 256-KiB slot of the caller's workspace as they stand (the ordinary pipelined epilogue with
//       `sc1` write-through stores), and once every wave's stores have drained (the store window's own counted wait + the interval barrier) ONE lane
//       sets the slot's flag with an `sc1` store;
//     * the tail piece -- the LAST thing workgroup w does, a whole tile or more after workgroup w - 8 wrote its slot -- polls the flag (`sc1` load),
//       runs an agent-scope acquire and starts its K loop from workgroup w - 8's slot instead of from zero: partial + the tail's products, a fixed order, so the
//       result does not depend on timing; its epilogue is the ordinary one.  The flag goes back to 0 at the end of the kernel.
//   A workgroup only ever waits for a LOWER-numbered workgroup's FIRST piece, which waits for nothing; the poll is bounded (an error word in the
//   workspace instead of a hang).  Every workgroup runs the same number of stages (+- 1): 108 instead of 128 at 432 tiles.
// Epilogues: fp32 rows (scale / bias / ReLU; 16-B stores) and chunk-major hi / lo planes.  Everything else (3-term products, interleaved outputs,
// fp32 A operands, other tiles) stays on cti_gemm_bf16x3.hip; gemm_nt_planes() routes.
#include "cti_common.h"

namespace cti {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int G16_ROWB = 64;                                         // bytes of K per stage row (32 bf16)
constexpr int G16_NS = 4;
constexpr int G16_EPI_LDS = 8 * 2 * 2 * 256;                         // per wave: 2 buffers x (bias | scale) x 64 floats
constexpr size_t G16_SK_FLAG_BYTES = 4096;                           // stream-K workspace: flag words in front of the partial slots

struct G16P {
    const char* Ah; const char* Bh;
    float* C; unsigned short* Ph; unsigned short* Pl;
    const float* scale; const float* bias;
    int64_t pitchA, pitchB;                    // chunk pitches in BYTES (A planes form); AROW: pitchA = the row stride of A in bytes
    int64_t rA1, rA2, rB1, rB2, kc2;           // batch strides in rows; split-K chunk offset per b2
    int64_t ldc_m, sC1, sC2, pitchP;           // pitchP in elements
    int64_t scale_bs, bias_bs;
    int nb2, M, N, Np, nk, scale_div, relu;
    int tiles_m, tiles_n, total_tiles;
    // stream-K (round 6; see "Stream-K" below): the LAST sk_tiles virtual tiles are cut into gridDim.x equal K ranges; 0 = every tile whole
    float* sk_part; unsigned* sk_flag;
    int sk_tiles;
};
typedef const __attribute__((address_space(4))) G16P G16P_K;
__device__ __forceinline__ const G16P_K* g16_kernarg() { return __builtin_bit_cast(const G16P_K*, __builtin_amdgcn_kernarg_segment_ptr()); }

__device__ __forceinline__ void g16_dma16(const char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void g16_dma4(const float* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
template <int V> __device__ __forceinline__ void g16_wait() { static_assert(V >= 0 && V < 64, "vmcnt has six bits"); asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(V) : "memory"); }
#define G16_BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ unsigned short g16_bf16_bits(float x) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x)); }
__device__ __forceinline__ float g16_bf16_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

enum { G16_EPI_F32 = 0, G16_EPI_PLANES = 1, G16_EPI_BF16 = 2 };

// Geometry: 8 waves as 2 (M) x 4 (N); a wave owns TMW x TNW MFMA tiles of 16 x 16: workgroup tile (32 TMW) x (64 TNW).  (8, 4) = 256 x 256;
// (9, 3) = 288 x 192 (9216 x 3072 outputs: 512 tiles = two full rounds of 256 workgroups instead of 432 tiles = 1.69 rounds of the square tile).
template <int TMW_, int TNW_, int NS_ = G16_NS>
struct G16Geo {
    // NS (round 6): slots of the LDS ring.  A stage's operands land ~2.4 us after their issue under load, so a tile's K loop runs at (stages in flight) / 2.4 us whatever the
    // tile's size: the big tiles are bound by the ring's BYTES (4 x 32 KiB), the small ones -- 16 or 24 KiB a stage -- take a deeper ring in the same LDS
    static constexpr int NS = NS_;
    static constexpr int TMW = TMW_, TNW = TNW_, BM = 32 * TMW_, BN = 64 * TNW_;
    static constexpr int PA = BM / 16, PB = BN / 16, P = PA + PB;                // LDS-DMA pieces (16 rows x 64 B) per stage
    static constexpr int UMAX = (P + 7) / 8, CQ = P / 8, CR = P % 8;             // waves < CR issue CQ + 1 pieces per stage, the others CQ
    static constexpr int STAGE = (BM + BN) * G16_ROWB;
    static constexpr int LDS = NS * STAGE + G16_EPI_LDS;
    static constexpr int NSTORE = TMW * TNW;                                     // fp32 / bf16-row epilogue: one store per accumulator tile
    static_assert(LDS <= 160 * 1024 && TNW <= 4, "ring + epilogue constants must fit the CU's LDS; a wave's columns fit one 64-lane dword piece");
    static_assert(NS >= 4 && NS <= 8 && (NS - 2) * (CQ + 1) + NSTORE < 64, "the store window must fit vmcnt's six bits");
};

// Stream-K range of workgroup w = x + 8 s: XCD x's chunk of the tile order holds total / 8 (+ 1 for x < total % 8) tiles, its first dp_tiles / 8 slots are whole
// rounds; the n remaining tiles are n * nk stages, of which slot s of the P / 8 takes an equal share [u0, u0 + len): tile ta from stage ka (la stages; 0 when the
// range starts on a tile boundary), nfull whole tiles, lb stages of the next
__device__ __forceinline__ void sk_range(int total, int dp_tiles, int nk, int w, int P, int& ta, int& ka, int& la, int& nfull, int& lb) {
    const int x = w & 7, s = w >> 3, S = P >> 3;
    const int n = (total >> 3) + (x < (total & 7) ? 1 : 0) - (dp_tiles >> 3);
    const int units = n * nk, base = units / S, extra = units - base * S;
    const int len = base + (s < extra ? 1 : 0), u0 = s * base + min(s, extra);
    ta = u0 / nk; ka = u0 - ta * nk; la = ka > 0 ? min(nk - ka, len) : 0;
    const int rem = len - la;
    nfull = rem / nk; lb = rem - nfull * nk;
}

template <int EPI, class G, bool AROW>
__global__ __launch_bounds__(512) void gemm16_planes_kernel(G16P p) {
    constexpr int NS = G::NS, STAGE = G::STAGE, BM = G::BM, BN = G::BN, TMW = G::TMW, TNW = G::TNW, UMAX = G::UMAX, NSTORE = G::NSTORE;
    constexpr bool SK = EPI != G16_EPI_PLANES;                       // stream-K pieces exist for the row epilogues only
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int total_tiles = p.total_tiles;
    if ((int)blockIdx.x >= total_tiles) return;
    const int nk = p.nk;
    // ---- this workgroup's pieces (tile, first stage, stages, kind): kind 0 = a whole tile or the K tail of one without a partial (ordinary epilogue),
    // 1 = K head (accumulators -> the workgroup's partial slot), 2 = K tail (starts from workgroup w - 1's slot)
    int npieces, total;
    {
        const int dp_tiles = total_tiles - p.sk_tiles, w = blockIdx.x, P = gridDim.x;
        const int ndp = w < dp_tiles ? (dp_tiles - 1 - w) / P + 1 : 0;
        npieces = ndp; total = ndp * nk;
        if (SK && p.sk_tiles > 0) {
            int ta, ka, la, nfull, lb;
            sk_range(total_tiles, dp_tiles, nk, w, P, ta, ka, la, nfull, lb);
            npieces += (lb > 0) + nfull + (la > 0); total += la + nfull * nk + lb;
        }
    }
    auto piece = [&](int pi, int& id, int& k0, int& kl, int& kind) {
        const G16P_K* q = g16_kernarg();
        asm volatile("" : "+s"(q));
        const int nkq = q->nk, dp_tiles = q->total_tiles - q->sk_tiles, w = blockIdx.x, P = gridDim.x;
        const int ndp = w < dp_tiles ? (dp_tiles - 1 - w) / P + 1 : 0;
        int ta = 0, ka = 0, la = 0, nfull = 0, lb = 0;
        if (SK && q->sk_tiles > 0) sk_range(q->total_tiles, dp_tiles, nkq, w, P, ta, ka, la, nfull, lb);
        // cut tile t of this XCD's chunk = slot dp_tiles / 8 + t of the chunk = virtual tile ((dp_tiles >> 3) + t) << 3 | x
        const int x = w & 7, first_full = ta + (ka > 0 ? 1 : 0);
        auto skid = [&](int t) { return (((dp_tiles >> 3) + t) << 3) | x; };
        int qi = pi - (lb > 0 ? 1 : 0);
        if (qi < 0) { id = skid(first_full + nfull); k0 = 0; kl = lb; kind = 1; }
        else if (qi < nfull) { id = skid(first_full + qi); k0 = 0; kl = nkq; kind = 0; }
        else if (qi < nfull + ndp) { id = w + (qi - nfull) * P; k0 = 0; kl = nkq; kind = 0; }
        else { id = skid(ta); k0 = ka; kl = la; kind = 2; }
    };
    const bool has_bias = p.bias != nullptr, has_scale = p.scale != nullptr;
    const bool hiw = G::CR > 0 && wid < G::CR;                       // this wave issues CQ + 1 pieces per stage (never, at 8 | P: folds away)

    // fragment addresses inside a slot: lane = (row lr of a 16-row MFMA tile, K group g of 8 elements = 16 B); unit' = g ^ f(row)
    const int lr = lane & 15, g = lane >> 4;
    const int fsw = (0 - (lr >> 2)) & 3;
    const int a_off = (wr * 16 * TMW + lr) * G16_ROWB + ((g ^ fsw) << 4);
    const int b_off = BM * G16_ROWB + (wc * 16 * TNW + lr) * G16_ROWB + ((g ^ fsw) << 4);
    char* const epi_lds = smem + NS * STAGE + wid * 1024;           // [buffer][bias 256 B | scale 256 B]

    // ---- DMA side: piece = 16 rows x 64 B of the stage's row list [A rows | B rows]; lane -> row (lane >> 2) of the piece, LDS unit' = lane & 3 <-
    // source unit c = (lane & 3) ^ f(row).  Planes: K chunk c >> 1 of the stage's two, half c & 1 of that chunk's 32 B; AROW: bytes 16 c of the
    // row's 64.  Wave w issues pieces w, w + 8, ...; piece g lands at g KiB of the slot.
    const int drow = lane >> 2, dc = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
    unsigned voff[UMAX]; bool pa[UMAX], pv[UMAX];
    const unsigned pitchA32 = (unsigned)p.pitchA, pitchB32 = (unsigned)p.pitchB;
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
        const int gp = wid + 8 * u;
        // (8 | PA and 8 | P: piece u of EVERY wave is an A piece for u < PA / 8 and exists -- compile-time, no scalar selects or branches per stage)
        pv[u] = G::P % 8 == 0 ? true : gp < G::P; pa[u] = G::PA % 8 == 0 ? u < G::PA / 8 : gp < G::PA;
        const int row = (pa[u] ? gp : gp - G::PA) * 16 + drow;
        if (pa[u] && AROW) voff[u] = 0;                              // per issue tile (rows past the matrix re-read its last row: no slack rows there)
        else voff[u] = (unsigned)(dc >> 1) * (pa[u] ? pitchA32 : pitchB32) + (unsigned)(row * 32 + (dc & 1) * 16);
    }
    const int64_t ksA = AROW ? 64 : 2 * p.pitchA, ksB = 2 * p.pitchB;
    // ---- issue side.  iss_left: stages of the open piece not yet issued (0: the next issue opens piece iss_pi); iss_rem: stages of the stream not yet issued.
    // Everything a stage's issue needs beyond its four LDS-DMA instructions is either rare (a piece opens) or advanced in the COMPUTE interval, between the
    // MFMAs: an instruction in the LOAD interval costs ~4 cycles of the interval that the partner wave group's MFMAs have to cover (round 6: the
    // per-stage scalar selects / branches / counters of the round-4 form were ~55 scalar instructions per wave and stage -- a third of the LOAD interval).
    int iss_pi = 0, iss_left = 0, iss_rem = total;
    const char* Ab = nullptr; const char* Bb = nullptr;              // wave-uniform: the open piece's operand origins at the next stage to issue
    auto issue_piece_open = [&]() {
        const G16P_K* q = g16_kernarg();
        asm volatile("" : "+s"(q));                                  // re-read once per piece instead of living in SGPRs across the K loop
        int z, tm, tn, id, k0, kind;
        piece(iss_pi, id, k0, iss_left, kind);
        tile_coords(id, q->total_tiles, q->tiles_m, q->tiles_n, z, tm, tn);
        const int b1 = z / q->nb2, b2 = z - b1 * q->nb2;
        if (AROW) {
            Ab = q->Ah + (b1 * q->rA1 + b2 * q->rA2 + (int64_t)tm * BM) * q->pitchA + b2 * q->kc2 * 32 + (int64_t)k0 * 64;
            const int last = q->M - 1 - tm * BM;
#pragma unroll
            for (int u = 0; u < UMAX; ++u)
                if (pa[u]) voff[u] = (unsigned)min((wid + 8 * u) * 16 + drow, last) * pitchA32 + (unsigned)(dc * 16);
        } else {
            Ab = q->Ah + (b1 * q->rA1 + b2 * q->rA2 + (int64_t)tm * BM) * 32 + (b2 * q->kc2 + 2 * (int64_t)k0) * q->pitchA;
        }
        Bb = q->Bh + (b1 * q->rB1 + b2 * q->rB2 + (int64_t)tn * BN) * 32 + (b2 * q->kc2 + 2 * (int64_t)k0) * q->pitchB;
        // the piece's epilogue constants ride in front of its first stage (buffer = the piece's parity; older than the stage's pieces: outside every counted window)
        const int n = min(tn * BN + wc * 16 * TNW + min(lane, 16 * TNW - 1), q->N - 1);
        if (has_bias) g16_dma4(q->bias + b1 * q->bias_bs + n, epi_lds + (iss_pi & 1) * 512);
        if (has_scale) g16_dma4(q->scale + b1 * q->scale_bs + n / q->scale_div, epi_lds + (iss_pi & 1) * 512 + 256);
        ++iss_pi;
    };
    auto issue_stage = [&](int slot_off) {
        char* sb = smem + slot_off + wid * 1024;
#pragma unroll
        for (int u = 0; u < UMAX; ++u)
            if (pv[u]) g16_dma16((pa[u] ? Ab : Bb) + voff[u], sb + u * 8192);
    };
    // all but this wave's youngest min(`stages`, NS - 2) stages of pieces (and, with `st`, the NSTORE unconditional stores issued behind them) have landed
    constexpr int D = NS - 2;                                        // stages in flight behind the one awaited
    auto wait_stages = [&](int stages, bool st) {
        constexpr int CH = G::CQ + 1, CL = G::CQ;
#define G16_W(n_) { if (hiw) { if (st) g16_wait<(n_) * CH + NSTORE>(); else g16_wait<(n_) * CH>(); } else { if (st) g16_wait<(n_) * CL + NSTORE>(); else g16_wait<(n_) * CL>(); } }
        if (stages >= D) G16_W(D)
        else if (D > 5 && stages == 5) G16_W(D > 5 ? 5 : 0)
        else if (D > 4 && stages == 4) G16_W(D > 4 ? 4 : 0)
        else if (D > 3 && stages == 3) G16_W(D > 3 ? 3 : 0)
        else if (D > 2 && stages == 2) G16_W(D > 2 ? 2 : 0)
        else if (stages == 1) G16_W(1)
        else G16_W(0)
#undef G16_W
    };

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (iss_rem > 0) {
            if (iss_left == 0) issue_piece_open();
            issue_stage(s * STAGE);
            Ab += ksA; Bb += ksB; --iss_left; --iss_rem;
        }
    wait_stages(total - 1, false);
    G16_BAR();
    if (wr == 1) G16_BAR();                                          // waves 4-7 run one interval behind

    f32x4 acc[TMW][TNW];
    bf16x8 fa[TMW], fb[TNW];
    // ---- compute side.  kleft: stages of the open piece (cpi: virtual tile cid, kind ckind) not yet multiplied; 0 in a LOAD interval = the piece ended with
    // the COMPUTE interval before: its epilogue and the next piece's accumulators (zeros, or a stream-K partial) are that LOAD interval's rare path
    int cpi = 0, kleft, cid, ckind;
    { int k0; piece(0, cid, k0, kleft, ckind); }
    int st_left = 0, sig_left = -1;                                  // LOAD intervals whose wait still has the store window; intervals until the partial's flag goes up (-1: none)
    bool had_tail = false;

    auto epilogue = [&](int z, int tm, int tn, int par, int kind) -> bool {
        const G16P_K* q = g16_kernarg();
        asm volatile("" : "+s"(q));
        if (SK && kind == 1) {
            // stream-K head piece: the accumulators as they stand -> this workgroup's slot, [wave][i][j][lane] x 16 B (1 KiB per store instruction),
            // write-through (`sc1`): the reader may sit on another XCD
            char* dst = reinterpret_cast<char*>(q->sk_part) + (size_t)blockIdx.x * (BM * BN * 4) + wid * (TMW * TNW * 1024) + lane * 16;
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j)
                    // (s_nop: a store of more than 8 bytes needs wait states before anything writes its data registers -- the compiler does not know this
                    // is a store, and it does reuse a dead accumulator as the next store's address)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + (i * TNW + j) * 1024), "v"(acc[i][j]) : "memory");
            return true;
        }
        const int pM = q->M, pN = q->N;
        const int b1 = z / q->nb2, b2 = z - b1 * q->nb2;
        const int m0 = tm * BM + wr * 16 * TMW, n0 = tn * BN + wc * 16 * TNW;
        const int ncols = EPI == G16_EPI_PLANES ? q->Np : pN;
        const bool full = (tm * BM + BM <= pM) && (tn * BN + BN <= pN);
        const bool relu = q->relu != 0;
        // the wave's bias / scale values: lane holds columns 16 j + 4 g + (0 .. 3)
        f32x4 bi[TNW], sc[TNW];
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            bi[j] = has_bias ? *reinterpret_cast<const f32x4*>(epi_lds + par * 512 + (j * 16 + 4 * g) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            sc[j] = has_scale ? *reinterpret_cast<const f32x4*>(epi_lds + par * 512 + 256 + (j * 16 + 4 * g) * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
        if (EPI == G16_EPI_F32 || EPI == G16_EPI_BF16) {
            // rows of fp32 (16-B stores) or bf16 (8-B stores: the next layer's A operand as it stands)
            constexpr int ESZ = EPI == G16_EPI_F32 ? 4 : 2;
            char* cp = reinterpret_cast<char*>(q->C) + (b1 * q->sC1 + b2 * q->sC2 + (int64_t)(m0 + lr) * q->ldc_m + n0 + 4 * g) * ESZ;
            const int64_t rstep = 16 * q->ldc_m * ESZ;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int m = m0 + i * 16 + lr;
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    f32x4 x = acc[i][j] * sc[j] + bi[j];
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = relu_nan(x[e]);
                    }
                    const int n = n0 + j * 16 + 4 * g;
                    const bool whole = full || (m < pM && n + 4 <= pN);
                    if (EPI == G16_EPI_F32) {
                        if (whole) *reinterpret_cast<f32x4*>(cp + j * 64) = x;
                        else if (m < pM) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (n + e < pN) reinterpret_cast<float*>(cp + j * 64)[e] = x[e];
                        }
                    } else {
                        const u32x2 hv = {g16_bf16_bits(x[0]) | ((unsigned)g16_bf16_bits(x[1]) << 16), g16_bf16_bits(x[2]) | ((unsigned)g16_bf16_bits(x[3]) << 16)};
                        if (whole) *reinterpret_cast<u32x2*>(cp + j * 32) = hv;
                        else if (m < pM) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (n + e < pN) reinterpret_cast<unsigned short*>(cp + j * 32)[e] = g16_bf16_bits(x[e]);
                        }
                    }
                }
                cp += rstep;
            }
        } else {
            // chunk-major planes: element (row, n) at (n >> 4) * pitchP + row * 16 + (n & 15); a lane's 4 columns are 8 B of one chunk row
            const int64_t prow0 = b1 * q->sC1 + b2 * q->sC2 + m0 + lr;
            const int64_t pitchP = q->pitchP;
            unsigned short* const Ph = q->Ph; unsigned short* const Pl = q->Pl;
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                const int m = m0 + i * 16 + lr;
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    f32x4 x = acc[i][j] * sc[j] + bi[j];
                    const int n = n0 + j * 16 + 4 * g;
                    unsigned short hb[4], lb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = relu ? relu_nan(x[e]) : x[e];
                        if (n + e >= pN) v = 0.f;                    // columns N .. Np - 1 of the planes are zero
                        hb[e] = g16_bf16_bits(v); lb[e] = g16_bf16_bits(v - g16_bf16_f32(hb[e]));
                    }
                    const int64_t o = (int64_t)(n >> 4) * pitchP + (prow0 + i * 16) * 16 + (n & 15);
                    const u32x2 hv = {hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16)};
                    const u32x2 lv = {lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16)};
                    if (full || (m < pM && n < ncols)) {
                        *reinterpret_cast<u32x2*>(Ph + o) = hv;
                        if (Pl) *reinterpret_cast<u32x2*>(Pl + o) = lv;
                    }
                }
            }
        }
        return full;
    };

#pragma unroll
    for (int u = 0; u < TMW; ++u)
#pragma unroll
        for (int v = 0; v < TNW; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a piece ended / the partial's flag is due / the ring is running out: the LOAD interval's rare path (everything else in it is unconditional)
    auto finish_piece = [&]() -> bool {
        const G16P_K* q = g16_kernarg();
        asm volatile("" : "+s"(q));
        int z, tm, tn;
        tile_coords(cid, q->total_tiles, q->tiles_m, q->tiles_n, z, tm, tn);
        return epilogue(z, tm, tn, cpi & 1, ckind);
    };
    int slot_rd = 0, slot_wr = (NS - 1) * STAGE, rem = total - 2;    // byte offsets of the slot read / filled; stages issued behind the one this interval's wait awaits
    bool evt = rem < D;
    for (int i = 0; i < total; ++i) {
        // ================= LOAD interval =================
        {
            const char* s = smem + slot_rd;
#pragma unroll
            for (int u = 0; u < TMW; ++u) fa[u] = *reinterpret_cast<const bf16x8*>(s + a_off + u * 1024);
#pragma unroll
            for (int u = 0; u < TNW; ++u) fb[u] = *reinterpret_cast<const bf16x8*>(s + b_off + u * 1024);
        }
        if (iss_rem > 0) {
            if (iss_left == 0) issue_piece_open();
            issue_stage(slot_wr);
        }
        if (evt) {
            if (kleft == 0) {
                const bool full = finish_piece();
                // (planes epilogue: up to 64 stores + the pieces exceed vmcnt's six bits: no store window there)
                st_left = (EPI != G16_EPI_PLANES && full && nk > NS - 2) ? NS - 1 : 0;
                if (SK && ckind == 1) sig_left = NS;                 // the slot's stores leave every wave's window NS - 1 waits on; one barrier more for the later wave group
                int k0;
                piece(++cpi, cid, k0, kleft, ckind);
                if (SK && ckind == 2) {
                    // stream-K tail piece: the K loop starts from workgroup w - 8's partial (flag poll: bounded; acquire: this CU's L1 may hold the slot's
                    // lines of an earlier launch)
                    const G16P_K* q = g16_kernarg();
                    asm volatile("" : "+s"(q));
                    const unsigned* fl = q->sk_flag + (blockIdx.x - 8);
                    int spins = 0;
                    // (readfirstlane: the loaded word is wave-uniform, and told so the loop -- and every counter behind it -- stays on the scalar unit)
                    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                        __builtin_amdgcn_s_sleep(16);
                        if (++spins > (1 << 20)) {
                            if (lane == 0) __hip_atomic_store(q->sk_flag + gridDim.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    const char* src = reinterpret_cast<const char*>(q->sk_part) + (size_t)(blockIdx.x - 8) * (BM * BN * 4) + wid * (TMW * TNW * 1024) + lane * 16;
                    // (inline asm: loads the compiler knew of would make it drain vmcnt -- the whole ring -- in front of EVERY interval's first MFMA)
#pragma unroll
                    for (int u = 0; u < TMW; ++u)
#pragma unroll
                        for (int v = 0; v < TNW; ++v) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(acc[u][v]) : "v"(src + (u * TNW + v) * 1024) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < TMW; ++u)
#pragma unroll
                        for (int v = 0; v < TNW; ++v) asm volatile("" : "+v"(acc[u][v]));     // the accumulators are defined HERE, behind the wait
                    had_tail = true;
                } else {
#pragma unroll
                    for (int u = 0; u < TMW; ++u)
#pragma unroll
                        for (int v = 0; v < TNW; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            if (SK && sig_left == 0) {
                // every wave's slot stores have drained and a barrier has passed since: one lane of the LATER wave group raises the flag
                sig_left = -1;
                if (wid == 4 && lane == 0) {
                    const G16P_K* q = g16_kernarg();
                    __hip_atomic_store(q->sk_flag + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // (readfirstlane: what the rare path leaves re-enters the loop on the scalar unit; without it the compiler keeps the counters in vector registers
            // and turns the loop's uniform branches into exec-mask sequences)
            kleft = __builtin_amdgcn_readfirstlane(kleft); ckind = __builtin_amdgcn_readfirstlane(ckind); cid = __builtin_amdgcn_readfirstlane(cid);
            cpi = __builtin_amdgcn_readfirstlane(cpi); st_left = __builtin_amdgcn_readfirstlane(st_left); sig_left = __builtin_amdgcn_readfirstlane(sig_left);
            wait_stages(rem, st_left > 0);
        } else {
            wait_stages(D, st_left > 0);
        }
        G16_BAR();
        // ================= COMPUTE interval: the MFMAs, and between them the scalar bookkeeping of the NEXT interval =================
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < TMW; ++u)
#pragma unroll
            for (int v = 0; v < TNW; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[v], fa[u], acc[u][v], 0, 0, 0);
        slot_rd = slot_rd == (NS - 1) * STAGE ? 0 : slot_rd + STAGE;
        slot_wr = slot_wr == (NS - 1) * STAGE ? 0 : slot_wr + STAGE;
        Ab += ksA; Bb += ksB; --iss_left; --iss_rem;                 // (as if this interval issued a stage: past the end of the stream nobody looks)
        --kleft; --rem;
        st_left = st_left > 0 ? st_left - 1 : 0;
        sig_left = sig_left > 0 ? sig_left - 1 : sig_left;
        evt = kleft == 0 || rem < D || sig_left == 0;
        // one bookkeeping instruction behind each MFMA: issued while the matrix pipe is busy with it, instead of in a row in front of the first / behind the last
#pragma unroll
        for (int k = 0; k < TMW * TNW; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x006, 1, 0); }
        __builtin_amdgcn_s_setprio(0);
        G16_BAR();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    (void)finish_piece();                                            // the last piece ended with the last COMPUTE interval
    if (wr == 0) G16_BAR();
    if (SK && (sig_left >= 0 || had_tail)) {                         // (workgroup-uniform) both wave groups have run every interval: plain barriers from here
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G16_BAR();
        if (wid == 0 && lane == 0) {
            const G16P_K* q = g16_kernarg();
            if (sig_left >= 0) __hip_atomic_store(q->sk_flag + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // a head piece within NS intervals of the end
            if (had_tail) __hip_atomic_store(q->sk_flag + (blockIdx.x - 8), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // consumed: ready for the next launch
        }
    }
}

// Stream-K plan for T tiles of nk stages on P workgroups (the kernel's header): the number of tiles cut (0 = every tile whole).
// Where it pays (measured, profiles/r06_gemm16_stream_k.txt): the cut gives every workgroup the same number of stages, but a cut tile's contributors walk K out
// of step with the workgroups that share its operand rows, so the XCD's L2 no longer serves one fetch of a K slice to all of them and the cut rounds run on the
// Infinity Cache's bandwidth (864 MB of operands for 432 tiles cut = ~9.5 TB/s at the uncut rate).  With three or more rounds the whole rounds in front
// still run in step and the cut is a net gain (9216 x 8192 x 2048: 282 -> 272 us); with one or two it is a loss (9216 x 3072 x 2048: 111 -> 123 us) and is left off
// unless forced (cti_set_tuning(CTI_TUNE_GEMM16_SK, 1): the tests).
int g16_sk_plan(long long T, int P, int nk) {
    const int mode = tuning_gemm16_sk();
    if (mode == 0 || T <= P || T % P == 0 || (P & 7) || nk < 16) return 0;
    const long long rounds = T / P, tsk = T - (rounds - 1) * P;
    if (mode < 0 && rounds < 3) return 0;
    const long long n_max = (T >> 3) + ((T & 7) ? 1 : 0) - ((rounds - 1) * P >> 3), S = P >> 3;     // the fullest XCD chunk's cut tiles, on S workgroups
    if (2ll * nk - (n_max * nk + S - 1) / S < 6) return 0;     // whole tiles: (rounds + 1) nk stages on the longest workgroup; cut: (rounds - 1) nk + ceil(n nk / S).  < 6 stages to gain: not worth a partial's round trip
    return (int)tsk;
}

template <int EPI, class G, bool AROW>
int g16_launch(G16P& p, int nb, int ncols, void* sk_ws, size_t sk_ws_bytes, hipStream_t st) {
    p.tiles_m = (p.M + G::BM - 1) / G::BM; p.tiles_n = (ncols + G::BN - 1) / G::BN;
    const long long total = (long long)nb * p.tiles_m * p.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return fail(CTI_E_SHAPE, "gemm16_planes: %lld tiles", total);
    p.total_tiles = (int)total;
    void (*kern)(G16P) = gemm16_planes_kernel<EPI, G, AROW>;
    static thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return fail((int)e, "gemm16_planes: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev = dev;
    }
    static thread_local int n_cu = 0;
    if (n_cu == 0) { (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    // CTI_GEMM16_LEAVE=n (experiment): the persistent launch takes n compute units fewer, so that another stream's chain of small kernels keeps running beside it
    static const int leave = [] { const char* e = getenv("CTI_GEMM16_LEAVE"); return e ? atoi(e) : 0; }();
    const int cus = (leave > 0 && leave < n_cu && total > n_cu - leave) ? n_cu - leave : n_cu;
    const long long grid = total < cus ? total : cus;
    // stream-K (EPI rows only; the caller's workspace = [flags: one word per workgroup + an error word, 4 KiB][one BM x BN fp32 slot per workgroup]); CTI_GEMM16_SK=0 = off
    static const bool sk_env = [] { const char* e = getenv("CTI_GEMM16_SK"); return !e || atoi(e) != 0; }();
    p.sk_tiles = 0; p.sk_part = nullptr; p.sk_flag = nullptr;
    if (EPI != G16_EPI_PLANES && sk_env && sk_ws && sk_ws_bytes >= G16_SK_FLAG_BYTES + (size_t)cus * G::BM * G::BN * 4 && (size_t)(cus + 1) * 4 <= G16_SK_FLAG_BYTES
        && (reinterpret_cast<uintptr_t>(sk_ws) & 255) == 0 && (p.sk_tiles = g16_sk_plan(total, cus, p.nk)) > 0) {
        p.sk_flag = static_cast<unsigned*>(sk_ws);
        p.sk_part = reinterpret_cast<float*>(static_cast<char*>(sk_ws) + G16_SK_FLAG_BYTES);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), G::LDS, st, p);
    return launch_status("gemm16_planes");
}

using G16Sq = G16Geo<8, 4>;                    // 256 x 256
using G16Wide = G16Geo<9, 3>;                  // 288 x 192
using G16Small = G16Geo<4, 2, 8>;              // 128 x 128, 8-slot ring (round 6: the model forwards' mid-size products, which the makespan model gives the small tiles)
using G16Tall = G16Geo<8, 2, 6>;               // 256 x 128, 6-slot ring

}  // namespace

// bytes of the stream-K workspace gemm16_planes() takes through PlaneGemmArgs::sk_ws (zeroed ONCE by the caller; the kernel leaves the flags at 0): the larger tile's slots
size_t gemm16_sk_workspace_bytes() {
    int dev = 0, n_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    return G16_SK_FLAG_BYTES + (size_t)n_cu * 256 * 256 * 4;
}

// Takes the plain-bf16 products gemm_nt_planes() would run on its 256 x 256 tile; false = not eligible (the caller's own kernel runs)
bool gemm16_eligible(const PlaneGemmArgs& a) {
    if (a.terms != 1 || a.Af || !(a.epi == 0 || a.epi == 1 || a.epi == 5) || a.ksplit > 1) return false;
    if (a.Kp % 32 != 0 || a.Kp <= 0) return false;
    if ((a.bias || a.scale) && a.Kp / 32 < 4) return false;                 // the epilogue constants' double buffer assumes tiles of at least 4 stages
    const int64_t rowsA = a.rows_allocA, rowsB = a.rows_allocB;
    if (rowsB * 32 * (int64_t)(a.Kp / 16) >= (1ll << 32)) return false;      // 32-bit per-lane offsets inside a plane
    if (a.Abf) {
        if ((a.ldabf & 7) || (reinterpret_cast<uintptr_t>(a.Abf) & 15) || a.ldabf * 2 * 288 >= (1ll << 32)) return false;
    } else if (rowsA * 32 * (int64_t)(a.Kp / 16) >= (1ll << 32)) return false;
    if (a.epi == 0) {
        if (a.ldc_n != 1 || (a.ldc_m & 3) || (a.sC1 & 3) || (a.sC2 & 3) || (reinterpret_cast<uintptr_t>(a.C) & 15)) return false;
    } else if (a.epi == 5) {
        if (a.ldc_n != 1 || (a.ldc_m & 3) || (a.sC1 & 3) || (a.sC2 & 3) || (reinterpret_cast<uintptr_t>(a.C) & 7)) return false;
    } else {
        if (!a.Ph || (a.Np & 15) || a.Abf) return false;
    }
    return true;
}

int gemm16_planes(const PlaneGemmArgs& a, hipStream_t st, int cfg) {
    G16P p{};
    p.Ah = a.Abf ? reinterpret_cast<const char*>(a.Abf) : reinterpret_cast<const char*>(a.Ah); p.Bh = reinterpret_cast<const char*>(a.Bh);
    p.C = a.C; p.Ph = a.Ph; p.Pl = a.Pl; p.scale = a.scale; p.bias = a.bias;
    p.pitchA = a.Abf ? a.ldabf * 2 : a.rows_allocA * 32; p.pitchB = a.rows_allocB * 32;
    p.rA1 = a.rA1; p.rA2 = a.rA2; p.rB1 = a.rB1; p.rB2 = a.rB2; p.kc2 = a.kc2;
    p.ldc_m = a.ldc_m; p.sC1 = a.sC1; p.sC2 = a.sC2; p.pitchP = a.rows_allocP * 16;
    p.scale_bs = a.scale_bs; p.bias_bs = a.bias_bs;
    p.nb2 = a.nb2 > 0 ? a.nb2 : 1; p.M = a.M; p.N = a.N; p.Np = a.Np; p.nk = a.Kp / 32; p.scale_div = a.scale_div > 0 ? a.scale_div : 1; p.relu = a.relu;
    const int ncols = a.epi == 1 ? a.Np : a.N;
    const int nb = a.nb1 * p.nb2;
    // cfg 0 / 1 (round 6): the 128 x 128 and 256 x 128 tiles of gemm_nt_planes()'s makespan model on THIS kernel -- the same two-group loop, 2 or 3 LDS-DMA pieces per wave
    // and stage.  The 128 x 128 planes kernel of cti_gemm_bf16x3.hip took ~0.6 us per 32-deep stage with one product per pair (a barrier round per stage for 4
    // MFMAs a wave): 3072 x 1024 x 1024 x 2 batches 38 us.
    if (cfg == 0 || cfg == 1) {
#define G16_GO_S(EPI) (cfg == 0 ? (a.Abf ? g16_launch<EPI, G16Small, true>(p, nb, ncols, nullptr, 0, st) : g16_launch<EPI, G16Small, false>(p, nb, ncols, nullptr, 0, st)) \
                                : (a.Abf ? g16_launch<EPI, G16Tall, true>(p, nb, ncols, nullptr, 0, st) : g16_launch<EPI, G16Tall, false>(p, nb, ncols, nullptr, 0, st)))
        if (a.epi == 1) return cfg == 0 ? g16_launch<G16_EPI_PLANES, G16Small, false>(p, nb, ncols, nullptr, 0, st) : g16_launch<G16_EPI_PLANES, G16Tall, false>(p, nb, ncols, nullptr, 0, st);
        if (a.epi == 5) return G16_GO_S(G16_EPI_BF16);
        return G16_GO_S(G16_EPI_F32);
#undef G16_GO_S
    }
    if (a.epi == 1) return g16_launch<G16_EPI_PLANES, G16Sq, false>(p, nb, ncols, nullptr, 0, st);
    // Tile: 256 x 256.  The 288 x 192 geometry (CTI_GEMM16_TILE=1; 9216 x 3072 outputs = 512 tiles = two FULL rounds of the 256 workgroups instead of 1.69)
    // is built and under test but measured no faster where its rounds are fewer (126.0 vs 128.5 us at 9216 x 3072 x 2048) and 20 % slower where they are equal
    // (539 vs 451 us at 9216 x 11264 x 2048): its COMPUTE interval is 27 MFMAs = 432 cycles against a LOAD interval of ~480 (12 fragment reads, 3-4 DMA
    // pieces at ~50 cycles of issue each, the LDS round trip): the square tile's 512-cycle COMPUTE interval is what just covers the LOAD interval.
    static const bool wide_env = [] { const char* e = getenv("CTI_GEMM16_TILE"); return e && atoi(e) == 1; }();
    //  the planes form of the A operand reads up to BM - 1 rows past M and the planes carry PLANE_SLACK_ROWS = 256 of slack: the 288-row tile is
    // for the row-major A operand only, whose DMA clamps its row index
    const bool wide = wide_env && a.Abf != nullptr;
#define G16_GO(EPI, GEO) (a.Abf ? g16_launch<EPI, GEO, true>(p, nb, ncols, a.sk_ws, a.sk_ws_bytes, st) : g16_launch<EPI, GEO, false>(p, nb, ncols, a.sk_ws, a.sk_ws_bytes, st))
    if (a.epi == 5) return wide ? G16_GO(G16_EPI_BF16, G16Wide) : G16_GO(G16_EPI_BF16, G16Sq);
    return wide ? G16_GO(G16_EPI_F32, G16Wide) : G16_GO(G16_EPI_F32, G16Sq);
#undef G16_GO
}

}  // namespace cti
