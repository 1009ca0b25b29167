// cti_gru.hip -- the question / answer GRU (reference src/language_model.py:57-61,91-96: nn.GRU(in, H, 1, batch_first=True) from a
// zero state), forward over all steps and back-propagation through time, as ONE library call each: the time loop runs on the
// host side of the C ABI, so a step costs two launches and no Python.
//
// Forward.  The input projection of every step is one plane GEMM (gi = x W_ih^T + b_ih).  The recurrent weights are split into
// bf16 hi/lo planes ONCE; a step is  (1) the skinny GEMM h_{t-1} W_hh^T as split-K partials (M = batch: K ranges as extra
// workgroups, cti_common.h plan_ksplit), (2) the gate kernel, which sums the partials, adds b_hh, applies the gates, writes
// h_t to out[:, t] and -- as bf16 hi/lo planes -- straight into the A operand of the next step's GEMM (no split pass).
// Backward mirrors it with W_hh^T planes: the gate-gradient kernel writes dgh_t as fp32 (for the weight gradients) and as planes
// (the A operand of dh_{t-1} += dgh_t W_hh).
#include "cti_common.h"

namespace cti {
namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ unsigned short bf16b(float x) { return __builtin_bit_cast(unsigned short, static_cast<__bf16>(x)); }
__device__ __forceinline__ float bf16f(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// The gates of one unit (order r, z, n; g* = the recurrent side incl. b_hh, gi* = the input side incl. b_ih).  The two multiply-adds are EXPLICIT fmas so that every
// kernel that calls this rounds alike whatever its surroundings let the compiler contract: the persistent form is tested bit for bit against the per-step form.
__device__ __forceinline__ float gru_gates(float gir, float giz, float gin, float g0, float g1, float g2, float hp, float& r, float& z, float& n) {
    r = sigm(gir + g0); z = sigm(giz + g1);
    n = tanhf(__builtin_fmaf(r, g2, gin));
    return __builtin_fmaf(z, hp, (1.f - z) * n);
}

// element (row, col) of a chunk-major plane pair: (col >> 4) * pitch + row * 16 + (col & 15)
__device__ __forceinline__ void store_planes(unsigned short* hi, unsigned short* lo, int64_t pitch, int row, int col, float x) {
    const int64_t o = (int64_t)(col >> 4) * pitch + (int64_t)row * 16 + (col & 15);
    const unsigned short h = bf16b(x);
    hi[o] = h;
    lo[o] = bf16b(x - bf16f(h));
}

// gh = sum_s part[s][b][3H] + b_hh;  gate order (r, z, n).  save: (B,5,H) = r, z, n, gh_n, h_t.
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ gi, int64_t ld_gi, const float* __restrict__ part, int S,
                                                           const float* __restrict__ b_hh, const float* __restrict__ hprev,
                                                           float* __restrict__ out, int64_t ld_out, float* __restrict__ h_tm,
                                                           float* __restrict__ save, unsigned short* __restrict__ ph,
                                                           unsigned short* __restrict__ pl, int64_t pitch, int B, int H) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * H) return;
    const int b = (int)(idx / H), j = (int)(idx % H);
    float g0 = b_hh[j], g1 = b_hh[H + j], g2 = b_hh[2 * H + j];
    const int64_t BN = (int64_t)B * 3 * H;
    for (int s = 0; s < S; ++s) {
        const float* p = part + s * BN + (int64_t)b * 3 * H;
        g0 += p[j]; g1 += p[H + j]; g2 += p[2 * H + j];
    }
    const float* gib = gi + (int64_t)b * ld_gi;
    const float hp = hprev ? hprev[idx] : 0.f;
    float r, z, n;
    const float h = gru_gates(gib[j], gib[H + j], gib[2 * H + j], g0, g1, g2, hp, r, z, n);
    out[(int64_t)b * ld_out + j] = h;
    if (h_tm) h_tm[idx] = h;
    if (save) {
        float* sv = save + (int64_t)b * 5 * H;
        sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = g2; sv[4 * H + j] = h;
    }
    if (ph) store_planes(ph, pl, pitch, b, j, h);
}

// dh = dout[b,t] + carry[b] + sum_s part[s][b];  writes dgi (row stride ld_dgi), dgh (fp32, contiguous (B,3H)), carry' = dh * z,
// and dgh as planes (rows b, columns 3H) for the next GEMM.
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* __restrict__ dout, int64_t ld_do, const float* __restrict__ carry_in,
                                                           const float* __restrict__ part, int S, const float* __restrict__ save,
                                                           const float* __restrict__ save_prev, float* __restrict__ dgi, int64_t ld_dgi,
                                                           float* __restrict__ dgh, float* __restrict__ carry_out,
                                                           unsigned short* __restrict__ ph, unsigned short* __restrict__ pl, int64_t pitch,
                                                           int B, int H) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * H) return;
    const int b = (int)(idx / H), j = (int)(idx % H);
    float dh = dout[(int64_t)b * ld_do + j];
    if (carry_in) dh += carry_in[idx];
    for (int s = 0; s < S; ++s) dh += part[(int64_t)s * B * H + idx];
    const float* sv = save + (int64_t)b * 5 * H;
    const float r = sv[j], z = sv[H + j], n = sv[2 * H + j], ghn = sv[3 * H + j];
    const float hp = save_prev ? save_prev[(int64_t)b * 5 * H + 4 * H + j] : 0.f;
    const float dan = dh * (1.f - z) * (1.f - n * n);
    const float daz = dh * (hp - n) * z * (1.f - z);
    const float dar = dan * ghn * r * (1.f - r);
    float* gi_ = dgi + (int64_t)b * ld_dgi;
    float* gh_ = dgh + (int64_t)b * 3 * H;
    gi_[j] = dar; gi_[H + j] = daz; gi_[2 * H + j] = dan;
    gh_[j] = dar; gh_[H + j] = daz; gh_[2 * H + j] = dan * r;
    carry_out[idx] = dh * z;
    if (ph) {
        store_planes(ph, pl, pitch, b, j, dar);
        store_planes(ph, pl, pitch, b, H + j, daz);
        store_planes(ph, pl, pitch, b, 2 * H + j, dan * r);
    }
}

// ---- fused forward step (bf16 modes): gh = h_{t-1} W_hh^T and the gates in ONE kernel, no split-K partials -------------------------------
// Tile: 64 batch rows x 16 hidden units x the 3 gates, the whole K = H: 4 waves, wave w owns rows 16w..16w+15 and the three 16 x 16 gate tiles
// of the workgroup's 16 units (v_mfma_f32_16x16x32_bf16), so r, z, n of a unit meet in one lane and the gate arithmetic is the epilogue.
// Operands: the chunk-major hi/lo planes of h_{t-1} (written by the previous step's epilogue) and of W_hh (split once), staged through a 4-slot
// LDS ring of 32-deep K steps by LDS-DMA with counted vmcnt (16 pieces of 1 KiB per slot: 4 per wave).  B = 256, H = 1024: 256 workgroups,
// each streams 448 KiB through LDS -- the floor of a step is that 115 MB of L2 -> LDS traffic (~9 us); the unfused form took a split-K GEMM
// (~19 us) + a partial-summing gate kernel (~5 us).
typedef float g_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
#ifndef CTI_GF_KPB
#define CTI_GF_KPB 1
#endif
constexpr int GF_KPB = CTI_GF_KPB;                                   // 32-deep K sub-steps per barrier of the fused step kernel (ring: GF_NST slots of GF_KPB sub-slots)
#ifndef CTI_GF_NST
#define CTI_GF_NST 4
#endif
constexpr int GF_NST = CTI_GF_NST, GF_SLOT = 16384;                         // slot: A [plane][chunk][64 rows][32 B] = 8 KiB, B likewise (48 of 64 rows used)
// -DCTI_GF_COMPACT=1 (round-5 experiment): the plain-bf16 mode stages the hi planes only -- 8 KiB of a slot -- in a ring of THREE such slots = 24 KiB, small enough to be
// co-resident with a workgroup of the 288 x 192 projection GEMM (128 KiB of LDS, 2 x 224 registers per SIMD lane): see DESIGN.md, round 5, VERDICT r4 #6.
#ifndef CTI_GF_COMPACT
#define CTI_GF_COMPACT 0
#endif
template <int TERMS> struct GfRing {
    static constexpr bool compact = CTI_GF_COMPACT && TERMS == 1;
    static constexpr int NST = compact ? 3 : GF_NST, SLOT = compact ? 8192 : GF_SLOT, OFF_B = compact ? 4096 : 8192;
};
template <int N> __device__ __forceinline__ void gf_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KPB (round 3 experiment): 32-deep K sub-steps per barrier (a slot holds KPB sub-slots of the same layout).  Measured on the MC model forward (16 GRU steps
// on its critical path; alternating runs on one box): KPB 1 / 2 / 4 (2-slot ring) / 2 (3-slot ring) = 0.913-0.923 / 0.923-0.927 / 0.927-0.932 / 0.922-0.924 ms:
// the step kernel (11.7 us at B = 256, H = 1 024) is not bound by its 32 barrier rounds; default stays 1.
#ifndef CTI_GF_DBG
#define CTI_GF_DBG 0      // debugging only (wrong results): 1 no DMA of the lo planes, 2 no lo-plane MFMAs
#endif
template <int TERMS, int KPB>
__global__ __launch_bounds__(256) void gru_step_fused_kernel(const unsigned short* __restrict__ Hh, const unsigned short* __restrict__ Hl, int64_t pitchH,
                                                             const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl, int64_t pitchW,
                                                             int nsteps, const float* __restrict__ gi, int64_t ld_gi, const float* __restrict__ b_hh,
                                                             const float* __restrict__ hprev, float* __restrict__ out, int64_t ld_out,
                                                             float* __restrict__ h_tm, float* __restrict__ save, unsigned short* __restrict__ ph,
                                                             unsigned short* __restrict__ pl, int64_t pitchP, int B, int H) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u0 = blockIdx.x * 16, row0 = blockIdx.y * 64;
    constexpr int NPL = TERMS == 3 ? 2 : 1;
    using R = GfRing<TERMS>;
    constexpr int NST = R::NST;
    // this wave's 4 DMA pieces per slot (2 with one plane): piece q of 16 = (operand, plane, chunk, half); a piece = 32 rows x 32 B of one plane-chunk.
    // A rows are contiguous in the plane; B rows are the three gates' 16-row groups: half 0 = gates r, z, half 1 = gate n (+ gate r again: unused)
    const unsigned short* src[4]; int ldso[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = wid * 4 + u, opnd = q >> 3, plane = (q >> 2) & 1, chunk = (q >> 1) & 1, half = q & 1;
        const int r32 = lane >> 1, c16 = lane & 1;                 // lane -> (row of the piece, 16-B half of the 32-B chunk row)
        const unsigned short* base; int64_t row;
        if (opnd == 0) { base = plane ? Hl : Hh; row = row0 + half * 32 + r32; base += (int64_t)chunk * pitchH; }
        else {
            const int gate = half == 0 ? (r32 >> 4) : (r32 < 16 ? 2 : 0);
            base = plane ? Wl : Wh; row = (int64_t)gate * H + u0 + (r32 & 15); base += (int64_t)chunk * pitchW;
        }
        src[u] = base + row * 16 + c16 * 8;
        ldso[u] = opnd * R::OFF_B + plane * 4096 + chunk * 2048 + half * 1024;       // (compact ring: plane 0 only)
    }
    const int64_t kstepH = 2 * pitchH, kstepW = 2 * pitchW;       // elements per 32-deep K step
    constexpr int SLOT = KPB * R::SLOT;
    auto issue = [&](int pos, int kg) {                            // kg: group of KPB sub-steps
#pragma unroll
        for (int sub = 0; sub < KPB; ++sub)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = wid * 4 + u;
            if ((NPL == 1 || (CTI_GF_DBG & 1)) && ((q >> 2) & 1)) continue;              // plain bf16: the lo planes are not staged
            const unsigned short* s_ = src[u] + (int64_t)(kg * KPB + sub) * ((q >> 3) ? kstepW : kstepH);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,
                                             (__attribute__((address_space(3))) void*)(gsm + pos * SLOT + sub * R::SLOT + ldso[u]), 16, 0, 0);
        }
    };
    constexpr int PER = 4 * KPB;                                   // DMA instructions per wave and slot (plain bf16: waves 0 and 2 issue them, waves 1 and 3 -- the lo planes -- none)
    nsteps /= KPB;                                                 // barrier groups (the launcher checks divisibility)
#ifndef CTI_GF_ABL
#define CTI_GF_ABL 0            // timing-only ablations (wrong results): 1 = no K loop (launch + epilogue only), 2 = K loop over a quarter of K
#endif
    if (CTI_GF_ABL & 1) nsteps = 0;
    if (CTI_GF_ABL & 2) nsteps /= 4;
    g_f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = g_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) if (i < nsteps) issue(i, i);
    // fragment addresses: lane l -> row (l & 15), k = 8 (l >> 4) .. + 7 of the 32-deep step = chunk (l >> 5), 16-B half (l >> 4) & 1
    const int fa = (lane >> 5) * 2048 + (wid * 16 + (lane & 15)) * 32 + ((lane >> 4) & 1) * 16;
    const int fb = R::OFF_B + (lane >> 5) * 2048 + (lane & 15) * 32 + ((lane >> 4) & 1) * 16;
    int pos = 0;
    for (int ks = 0; ks < nsteps; ++ks) {
        const int rem = nsteps - 1 - ks;
        if (rem >= NST - 2) gf_wait<(NST - 2) * PER>(); else if (rem == 1) gf_wait<PER>(); else gf_wait<0>();
        __builtin_amdgcn_s_barrier();
        if (ks + NST - 1 < nsteps) issue(pos == 0 ? NST - 1 : pos - 1, ks + NST - 1);
#pragma unroll
        for (int sub = 0; sub < KPB; ++sub) {
        const char* s = gsm + pos * SLOT + sub * R::SLOT;
        const g_bf16x8 ah = *reinterpret_cast<const g_bf16x8*>(s + fa);
        g_bf16x8 al = ah;
        if (TERMS == 3) al = *reinterpret_cast<const g_bf16x8*>(s + fa + 4096);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int go = g < 2 ? g * 512 : 1024;                 // gates r, z: rows 0-15, 16-31 of half 0; gate n: rows 0-15 of half 1
            const g_bf16x8 bh = *reinterpret_cast<const g_bf16x8*>(s + fb + go);
            if (TERMS == 3 && !(CTI_GF_DBG & 2)) {
                const g_bf16x8 bl = *reinterpret_cast<const g_bf16x8*>(s + fb + go + 4096);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[g], 0, 0, 0);
            }
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[g], 0, 0, 0);
        }
        }
        pos = pos == NST - 1 ? 0 : pos + 1;
    }
    // epilogue: C/D map of the 16x16 MFMA: column = lane & 15 (unit), row = 4 (lane >> 4) + reg
    const int j = u0 + (lane & 15);
    if (j >= H) return;
    const float bb0 = b_hh[j], bb1 = b_hh[H + j], bb2 = b_hh[2 * H + j];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int b = row0 + wid * 16 + (lane >> 4) * 4 + e;
        if (b >= B) continue;
        const float g0 = acc[0][e] + bb0, g1 = acc[1][e] + bb1, g2 = acc[2][e] + bb2;
        const float* gib = gi + (int64_t)b * ld_gi;
        const int64_t idx = (int64_t)b * H + j;
        const float hp = hprev ? hprev[idx] : 0.f;
        float r, z, n;
        const float h = gru_gates(gib[j], gib[H + j], gib[2 * H + j], g0, g1, g2, hp, r, z, n);
        out[(int64_t)b * ld_out + j] = h;
        if (h_tm) h_tm[idx] = h;
        if (save) {
            float* sv = save + (int64_t)b * 5 * H;
            sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = g2; sv[4 * H + j] = h;
        }
        if (ph) store_planes(ph, pl, pitchP, b, j, h);
    }
}

// ---- fused forward step without a barrier in its K loop (round 6) ----------------------------------------------------------------------------------
// The ring kernel above walks K = H in 32 rounds of (counted wait, barrier, 3-6 MFMAs per wave): ~6.5 us of its ~12 us are that loop, and the gate inputs are only
// fetched behind it.  Same tile here (64 batch rows x 16 hidden units x 3 gates per workgroup), but EIGHT waves, each with a contiguous eighth of K for which it
// computes the whole tile (4 row tiles x 3 gates = 12 accumulators): its A fragments (h_{t-1} planes) and B fragments (W_hh planes) come straight from global
// memory as 16-B loads, all of a batch in flight at once (28 loads), no LDS staging, no barrier; the eight partial tiles meet in LDS (two halves of 48 KiB, fixed
// order: deterministic), where thread (row, unit) finds the three gates of its output.  The gate inputs (gi, h_{t-1}, b_hh) are loaded BEFORE the K loop.
#ifndef CTI_GRU_KS_SB
#define CTI_GRU_KS_SB 2        // K steps of fragments in flight per batch in the plain-bf16 step kernel.  2: <= 128 registers, TWO workgroups per compute unit -- the steps of
                               // two GRUs (question / answer, BAN / CTI on sibling streams) share the chip instead of alternating: c4 1 400 -> 1 358 us, c3 unchanged; alone a step
                               // is ~1 us longer than with 4 (one more round trip): profiles/r06_gru_step_k_split.txt
#endif
template <int TERMS>
__global__ __launch_bounds__(512, (CTI_GRU_KS_SB == 2 ? 4 : 2)) void gru_step_ks_kernel(const unsigned short* __restrict__ Hh, const unsigned short* __restrict__ Hl, int64_t pitchH,
                                                          const unsigned short* __restrict__ Wh, const unsigned short* __restrict__ Wl, int64_t pitchW,
                                                          int nsteps, const float* __restrict__ gi, int64_t ld_gi, const float* __restrict__ b_hh,
                                                          const float* __restrict__ hprev, float* __restrict__ out, int64_t ld_out,
                                                          float* __restrict__ h_tm, float* __restrict__ save, unsigned short* __restrict__ ph,
                                                          unsigned short* __restrict__ pl, int64_t pitchP, int B, int H) {
    __shared__ __attribute__((aligned(16))) float part[8 * 6 * 4 * 64];      // [wave][tile = (m & 1) * 3 + gate][reg][lane]: 48 KiB, one half of the row tiles at a time
    const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int u0 = blockIdx.x * 16, row0 = blockIdx.y * 64;
    // epilogue role: thread -> (row er of a half's 32, unit ej); its gate inputs are in flight while the K loop runs
    const int er = t >> 4, ej = t & 15, j = u0 + ej;
    float gv[2][3], hpv[2], bb[3];
    const bool jok = j < H;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int b = row0 + p * 32 + er;
        const bool ok = jok && b < B;
        const float* gib = gi + (int64_t)(ok ? b : 0) * ld_gi;
        gv[p][0] = ok ? gib[j] : 0.f; gv[p][1] = ok ? gib[H + j] : 0.f; gv[p][2] = ok ? gib[2 * H + j] : 0.f;
    }
    // fragments of K step ks: chunk 2 ks + (lane >> 5), row (lane & 15) of the tile, 16-B half (lane >> 4) & 1
    const int spw = (nsteps + 7) / 8, s0 = wid * spw, my = max(0, min(nsteps, s0 + spw) - s0);
    const unsigned short* ah = Hh + (int64_t)(lane >> 5) * pitchH + (int64_t)(row0 + (lane & 15)) * 16 + ((lane >> 4) & 1) * 8;
    const unsigned short* al = Hl + (int64_t)(lane >> 5) * pitchH + (int64_t)(row0 + (lane & 15)) * 16 + ((lane >> 4) & 1) * 8;
    const int wrow = min(u0 + (lane & 15), H - 1);
    const unsigned short* bh = Wh + (int64_t)(lane >> 5) * pitchW + (int64_t)wrow * 16 + ((lane >> 4) & 1) * 8;
    const unsigned short* bl = Wl + (int64_t)(lane >> 5) * pitchW + (int64_t)wrow * 16 + ((lane >> 4) & 1) * 8;
    g_f32x4 acc[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[m][g] = g_f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int SB = TERMS == 3 ? (CTI_GRU_KS_SB == 2 ? 1 : 2) : CTI_GRU_KS_SB;      // K steps whose fragments are in flight together (28 loads of 16 B per lane at the default)
    for (int sb = 0; sb < my; sb += SB) {
        g_bf16x8 fah[SB][4], fal[SB][4], fbh[SB][3], fbl[SB][3];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            if (sb + i < my) {
                const int64_t ka = (int64_t)(s0 + sb + i) * 2 * pitchH, kb = (int64_t)(s0 + sb + i) * 2 * pitchW;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    fah[i][m] = *reinterpret_cast<const g_bf16x8*>(ah + ka + m * 256);
                    if (TERMS == 3) fal[i][m] = *reinterpret_cast<const g_bf16x8*>(al + ka + m * 256);
                }
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    fbh[i][g] = *reinterpret_cast<const g_bf16x8*>(bh + kb + (int64_t)g * H * 16);
                    if (TERMS == 3) fbl[i][g] = *reinterpret_cast<const g_bf16x8*>(bl + kb + (int64_t)g * H * 16);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            if (sb + i < my) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        if (TERMS == 3) {
                            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i][m], fbh[i][g], acc[m][g], 0, 0, 0);
                            acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i][m], fbl[i][g], acc[m][g], 0, 0, 0);
                        }
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i][m], fbh[i][g], acc[m][g], 0, 0, 0);
                    }
            }
        }
    }
    // C/D map of the 16x16 MFMA: column = lane & 15 (unit), row = 4 (lane >> 4) + reg.  Two halves of the rows; thread (er, ej) sums the eight waves' partials of its
    // three gates in wave order and applies the gates
    bb[0] = jok ? b_hh[j] : 0.f; bb[1] = jok ? b_hh[H + j] : 0.f; bb[2] = jok ? b_hh[2 * H + j] : 0.f;      // (in flight across the first barrier below: three registers the K loop does not carry)
#pragma unroll
    for (int p = 0; p < 2; ++p) { const int b = row0 + p * 32 + er; hpv[p] = (jok && b < B && hprev) ? hprev[(int64_t)b * H + j] : 0.f; }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p) __syncthreads();
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) part[((wid * 6 + mm * 3 + g) * 4 + e) * 64 + lane] = acc[2 * p + mm][g][e];
        __syncthreads();
        const int mm = er >> 4, rr = er & 15;
        const float* pp = part + ((mm * 3) * 4 + (rr & 3)) * 64 + (rr >> 2) * 16 + ej;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { g0 += pp[w * 6 * 256]; g1 += pp[w * 6 * 256 + 256]; g2 += pp[w * 6 * 256 + 512]; }
        const int b = row0 + p * 32 + er;
        if (jok && b < B) {
            g0 += bb[0]; g1 += bb[1]; g2 += bb[2];
            float r, z, n;
            const float h = gru_gates(gv[p][0], gv[p][1], gv[p][2], g0, g1, g2, hpv[p], r, z, n);
            const int64_t idx = (int64_t)b * H + j;
            out[(int64_t)b * ld_out + j] = h;
            if (h_tm) h_tm[idx] = h;
            if (save) {
                float* sv = save + (int64_t)b * 5 * H;
                sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = g2; sv[4 * H + j] = h;
            }
            if (ph) store_planes(ph, pl, pitchP, b, j, h);
        }
    }
}

// ---- persistent forward (plain bf16, inference; round 6): ALL time steps in ONE launch ----------------------------------------------------------
// The per-step kernel above re-streams its slice of W_hh (96 KiB) through LDS every step and pays a kernel boundary per step: ~12 us a step at
// B = 256, H = 1 024 for 0.3 us of MFMA work.  Here a workgroup (64 batch rows x 16 hidden units, 4 waves) keeps its W_hh slice in LDS for the
// whole sequence -- [K step][chunk][gate * 16 + unit][32 B], 3 KiB per 32-deep K step -- and its own h_{t-1} values in registers; what crosses
// workgroups per step is h_t as bf16 (the chunk-major plane the step kernel also writes: 2 KiB per workgroup, 512 KiB in all), handed over by
// the recipe of MI355X_MICROARCH.md's hand-off table (third row), with NO agent-scope fence on either side (round 4's attempt paid ~7 us a step
// for a release + an acquire):
//   producer: the 16 x 16 tile of a wave goes through 512 B of LDS so that ONE `global_store_dwordx4 sc1` of 32 lanes writes four whole 128-B
//             lines; every storing wave waits vmcnt(0); workgroup barrier; ONE lane adds 1 to the row block's counter (agent-scope atomic);
//   consumer: ONE lane polls that counter with `sc1` loads until all gridDim.x workgroups of the row block have added t times; workgroup barrier;
//             every wave then reads its 16 rows x 1 024 k of h_{t-1} straight into MFMA A fragments with `buffer_load_dwordx4 sc1` (served by
//             the L2 / the memory side, never by a stale L1 line).
// Ping-pong planes: a workgroup that has passed the poll of step t knows every workgroup has finished READING h_{t-2} (they added after it), so
// step t may overwrite that plane.  Row blocks never wait for each other.  EVERY workgroup of a row block must be resident for the others to pass:
// the launcher refuses grids beyond the device's compute-unit count, the caller keeps two such launches from sharing the chip (ops.py: one stream
// per device carries them), and a poll that exceeds `spin_limit` rounds sets the error word and NaN-fills this workgroup's outputs instead of hanging.
typedef unsigned gp_u32x4 __attribute__((ext_vector_type(4)));
constexpr int GP_TR = 512;                                          // LDS bytes per wave for the tile transposition
constexpr int GP_SYNC_BYTES = 4096;                                 // counters: one 128-B line per row block (<= 31), then the error word's line
__global__ __launch_bounds__(256, 1) void gru_persistent_kernel(unsigned short* __restrict__ hp0, unsigned short* __restrict__ hp1, int64_t pitchH, size_t plane_bytes,
                                                                const unsigned short* __restrict__ Wh, int64_t pitchW, int nsteps,
                                                                const float* __restrict__ gi, int64_t ld_gi, const float* __restrict__ b_hh,
                                                                float* __restrict__ out, int64_t ld_out, int T, int H, unsigned* __restrict__ sync,
                                                                unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) char psm[];      // W_hh slice, then 4 x GP_TR
    __shared__ int dead;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int u0 = blockIdx.x * 16, row0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < nsteps * 192; i += 256) {          // 16-B units: (K step, chunk, row rr = gate * 16 + unit, half)
        const int hf = i & 1, rr = (i >> 1) % 48, c = ((i >> 1) / 48) & 1, ks = (i >> 1) / 96;
        const unsigned short* src = Wh + (int64_t)(2 * ks + c) * pitchW + ((int64_t)(rr >> 4) * H + u0 + (rr & 15)) * 16 + hf * 8;
        *reinterpret_cast<uint4*>(psm + ks * 3072 + c * 1536 + rr * 32 + hf * 16) = *reinterpret_cast<const uint4*>(src);
    }
    if (threadIdx.x == 0) dead = 0;
    __syncthreads();
    char* tr = psm + nsteps * 3072 + wid * GP_TR;
    const int fb = (lane >> 5) * 1536 + (lane & 15) * 32 + ((lane >> 4) & 1) * 16;
    // A fragment of K step ks: row (lane & 15) of this wave's 16, k = 8 (lane >> 4) .. + 7 = chunk 2 ks + (lane >> 5), 16-B half (lane >> 4) & 1
    const unsigned a_off = (unsigned)(((int64_t)(lane >> 5) * pitchH + (int64_t)(row0 + wid * 16 + (lane & 15)) * 16 + ((lane >> 4) & 1) * 8) * 2);
    const unsigned a_step = (unsigned)(4 * pitchH);                  // bytes per K step (two chunks)
    // this lane's 16 B of the wave's stored tile: row (lane >> 1), half (lane & 1) of chunk u0 / 16 (lanes < 32 store)
    const int64_t st_off = (int64_t)(u0 >> 4) * pitchH + (int64_t)(row0 + wid * 16 + (lane >> 1)) * 16 + (lane & 1) * 8;
    const int j = u0 + (lane & 15);
    const float bb0 = b_hh[j], bb1 = b_hh[H + j], bb2 = b_hh[2 * H + j];
    unsigned* ctr = sync + blockIdx.y * 32;
    float hprev[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        unsigned short* hw = (t & 1) ? hp1 : hp0;                   // h_t goes here; h_{t-1} is in the other plane
        const unsigned short* hr = (t & 1) ? hp0 : hp1;
        float gv[12];                                               // this step's input-side pre-activations: in flight across the poll
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* gib = gi + (int64_t)(row0 + wid * 16 + (lane >> 4) * 4 + e) * ld_gi + (int64_t)t * 3 * H;
            gv[3 * e] = gib[j]; gv[3 * e + 1] = gib[H + j]; gv[3 * e + 2] = gib[2 * H + j];
        }
        g_f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = g_f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            if (threadIdx.x == 0) {                                 // every workgroup of this row block has stored h_{t-1}
                const unsigned target = gridDim.x * (unsigned)t;
                unsigned spins = 0;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_limit) { __hip_atomic_fetch_or(sync + 32 * 31, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = 1; break; }
                }
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(hr), 0, (int)plane_bytes, 0x00020000);
            constexpr int NB = 32;                                  // K steps whose fragments are in flight together (all of them at H = 1 024: 128 registers)
            for (int k0 = 0; k0 < nsteps; k0 += NB) {
                gp_u32x4 af[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (k0 + i < nsteps) af[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_off + (unsigned)(k0 + i) * a_step, 0, 16 /* sc1 */);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (k0 + i >= nsteps) break;
                    const g_bf16x8 ah = __builtin_bit_cast(g_bf16x8, af[i]);
                    const char* wb = psm + (k0 + i) * 3072 + fb;
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const g_bf16x8 bh = *reinterpret_cast<const g_bf16x8*>(wb + g * 512);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[g], 0, 0, 0);
                    }
                }
            }
        }
        const bool poisoned = dead != 0;
        // C/D map of the 16x16 MFMA: column = lane & 15 (unit), row = 4 (lane >> 4) + e
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rl = (lane >> 4) * 4 + e;
            const float g0 = acc[0][e] + bb0, g1 = acc[1][e] + bb1, g2 = acc[2][e] + bb2;
            float r, z, n;
            float h = gru_gates(gv[3 * e], gv[3 * e + 1], gv[3 * e + 2], g0, g1, g2, hprev[e], r, z, n);
            if (poisoned) h = __builtin_nanf("");
            hprev[e] = h;
            out[(int64_t)(row0 + wid * 16 + rl) * ld_out + (int64_t)t * H + j] = h;
            *reinterpret_cast<unsigned short*>(tr + rl * 32 + (lane & 15) * 2) = bf16b(h);
        }
        if (t + 1 < T) {
            // the wave's own tile, transposed through its own 512 B of LDS (wave-private: no barrier), as whole lines
            if (lane < 32) {
                const gp_u32x4 v = *reinterpret_cast<const gp_u32x4*>(tr + lane * 16);
                unsigned short* dst = hw + st_off;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// zero fill as a kernel (a hipMemsetAsync captured into a hipGraph was not re-executed reliably on replay; see cti_attention.hip)
__global__ __launch_bounds__(256) void zero_u16_kernel(unsigned short* __restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0;
}
inline int zero_planes(unsigned short* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / sizeof(unsigned short);
    hipLaunchKernelGGL(zero_u16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
    return launch_status("zero_planes");
}

struct Carve {
    char* p; size_t used;
    template <class T> T* take(size_t bytes) { T* r = reinterpret_cast<T*>(p + used); used += al256(bytes); return r; }
};

// the recurrent GEMM of one step: C_part[s] = A[:, ks] B[:, ks]^T  (A: M rows, B: N rows, both planes of depth Kp), S = plan or 1
int step_gemm(const unsigned short* ah, const unsigned short* al, int64_t ra, const unsigned short* bh, const unsigned short* bl, int64_t rb,
              int M, int N, int Kp, int S, int terms, float* part, hipStream_t st) {
    PlaneGemmArgs g{};
    g.Ah = ah; g.Al = al; g.Bh = bh; g.Bl = bl; g.rows_allocA = ra; g.rows_allocB = rb;
    g.nb1 = 1; g.nb2 = S; g.kc2 = S > 1 ? Kp / S / 16 : 0;
    g.M = M; g.N = N; g.Kp = Kp / S; g.terms = terms; g.epi = 0;
    g.C = part; g.ldc_m = N; g.ldc_n = 1; g.sC2 = (int64_t)M * N;
    return gemm_nt_planes(g, st);
}

}  // namespace
}  // namespace cti

using namespace cti;

extern "C" {

size_t cti_gru_forward_workspace_bytes(int B, int T, int I, int H, int prec) {
    if (B <= 0 || T <= 0 || I <= 0 || H <= 0) return 0;
    const size_t f = sizeof(float);
    size_t n = al256(f * (size_t)B * T * 3 * H) + al256(f * (size_t)T * B * H);                 // gi, time-major h
    if (prec == CTI_PREC_F32) return n + al256(f * (size_t)B * 3 * H);
    const int S = plan_ksplit(B, 3 * H, planes_kp(H), 1);
    n += al256(f * (size_t)S * B * 3 * H);
    n += al256(planes_bytes((int64_t)B * T + PLANE_SLACK_ROWS, I)) + al256(planes_bytes(3 * (int64_t)H + PLANE_SLACK_ROWS, I));
    n += al256(planes_bytes(3 * (int64_t)H + PLANE_SLACK_ROWS, H)) + 2 * al256(planes_bytes((int64_t)B + PLANE_SLACK_ROWS, H));
    return n + al256(GP_SYNC_BYTES);                                                             // the persistent form's counters
}

// The persistent form (gru_persistent_kernel) takes a call when the caller has asked for it (CTI_TUNE_GRU_PERSISTENT = 1: it needs the whole chip, see the
// kernel's header), the arithmetic is plain bf16, no per-step state is saved for a backward pass, the tile grid divides the problem and every workgroup can
// be resident at once.
static bool gru_persistent_fits(int B, int H, int prec, const float* save) {
    if (tuning_gru_persistent() != 1 || prec != CTI_PREC_BF16 || save || B % 64 || H % 32 || H > 1024 || B / 64 > 31) return false;
    static thread_local int cus_dev = -1, cus = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (cus_dev != dev) { if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0; cus_dev = dev; }
    return (H / 16) * (B / 64) <= cus;
}

static int gru_forward_impl(const float* x, const void* x16, int64_t ldx16, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* out, float* save,
                            int B, int T, int I, int H, int prec, const void* w_ih_planes, const void* w_hh_planes, void* workspace, size_t workspace_bytes,
                            void* stream);

int cti_gru_forward(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* out, float* save,
                    int B, int T, int I, int H, int prec, const void* w_ih_planes, const void* w_hh_planes, void* workspace, size_t workspace_bytes,
                    void* stream) {
    CTI_REQUIRE_PTR(x);
    return gru_forward_impl(x, nullptr, 0, w_ih, w_hh, b_ih, b_hh, out, save, B, T, I, H, prec, w_ih_planes, w_hh_planes, workspace, workspace_bytes, stream);
}

// x as bf16 rows (B * T rows of pitch ldx = I rounded up to a multiple of 32, zero beyond I; 16-B aligned): the plain-bf16 mode's input-side product reads them as they
// stand (cti_embedding_fwd_bf16 writes them): no fp32 word vectors, no split pass
int cti_gru_forward_x16(const void* x_bf16, int64_t ldx, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* out, float* save,
                        int B, int T, int I, int H, int prec, const void* w_ih_planes, const void* w_hh_planes, void* workspace, size_t workspace_bytes,
                        void* stream) {
    CTI_REQUIRE_PTR(x_bf16);
    CTI_REQUIRE(prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_gru_forward_x16: prec=%d (bf16 word vectors are the plain-bf16 mode's)", prec);
    CTI_REQUIRE(I > 0 && ldx == planes_kp(I) && (reinterpret_cast<uintptr_t>(x_bf16) & 15) == 0, CTI_E_ALIGN,
                "cti_gru_forward_x16: rows of pitch %d (I = %d rounded up to 32) and a 16-B aligned base are required (ldx=%lld)", planes_kp(I > 0 ? I : 1), I, (long long)ldx);
    return gru_forward_impl(nullptr, x_bf16, ldx, w_ih, w_hh, b_ih, b_hh, out, save, B, T, I, H, prec, w_ih_planes, w_hh_planes, workspace, workspace_bytes, stream);
}

static int gru_forward_impl(const float* x, const void* x16, int64_t ldx16, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, float* out, float* save,
                            int B, int T, int I, int H, int prec, const void* w_ih_planes, const void* w_hh_planes, void* workspace, size_t workspace_bytes,
                            void* stream) {
    CTI_REQUIRE_PTR(w_ih); CTI_REQUIRE_PTR(w_hh); CTI_REQUIRE_PTR(b_ih); CTI_REQUIRE_PTR(b_hh); CTI_REQUIRE_PTR(out);
    CTI_REQUIRE(B > 0 && T > 0 && I > 0 && H > 0 && (int64_t)B * T < (1ll << 31), CTI_E_SHAPE, "cti_gru_forward: B=%d T=%d I=%d H=%d", B, T, I, H);
    CTI_REQUIRE(prec == CTI_PREC_F32 || prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_gru_forward: prec=%d", prec);
    CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(workspace_bytes >= cti_gru_forward_workspace_bytes(B, T, I, H, prec), CTI_E_WORKSPACE, "cti_gru_forward: workspace too small");
    hipStream_t st = as_stream(stream);
    Carve ws{static_cast<char*>(workspace), 0};
    const size_t f = sizeof(float);
    const int H3 = 3 * H;
    float* gi = ws.take<float>(f * (size_t)B * T * H3);                     // (B, T, 3H)
    float* h_tm = ws.take<float>(f * (size_t)T * B * H);                    // (T, B, H): contiguous h_{t-1} for the fp32 GEMM
    const unsigned nblk = (unsigned)(((int64_t)B * H + 255) / 256);
    if (prec == CTI_PREC_F32) {
        float* gh = ws.take<float>(f * (size_t)B * H3);
        GemmP p{};
        p.A = x; p.B = w_ih; p.C = gi; p.lda = I; p.ldb = I; p.ldc_m = H3; p.ldc_n = 1; p.nb1 = 1; p.nb2 = 1;
        p.M = B * T; p.N = H3; p.K = I; p.scale_div = 1; p.bias = b_ih;
        int rc = gemm_nt_f32(p, st); if (rc) return rc;
        for (int t = 0; t < T; ++t) {
            if (t) {
                GemmP q{};
                q.A = h_tm + (size_t)(t - 1) * B * H; q.B = w_hh; q.C = gh; q.lda = H; q.ldb = H; q.ldc_m = H3; q.ldc_n = 1; q.nb1 = 1; q.nb2 = 1;
                q.M = B; q.N = H3; q.K = H; q.scale_div = 1;
                rc = gemm_nt_f32(q, st); if (rc) return rc;
            }
            hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(nblk), dim3(256), 0, st, gi + (size_t)t * H3, (int64_t)T * H3, gh, t ? 1 : 0, b_hh,
                               t ? h_tm + (size_t)(t - 1) * B * H : nullptr, out + (size_t)t * H, (int64_t)T * H, h_tm + (size_t)t * B * H,
                               save ? save + (size_t)t * B * 5 * H : nullptr, nullptr, nullptr, (int64_t)0, B, H);
            rc = launch_status("cti_gru_forward/step"); if (rc) return rc;
        }
        return CTI_OK;
    }
    const int terms = prec == CTI_PREC_BF16X3 ? 3 : 1;
    const int KpI = planes_kp(I), KpH = planes_kp(H);
    const int S = plan_ksplit(B, H3, KpH, 1);
    float* part = ws.take<float>(f * (size_t)S * B * H3);
    const int64_t rx = (int64_t)B * T + PLANE_SLACK_ROWS, rw = (int64_t)H3 + PLANE_SLACK_ROWS, rh = (int64_t)B + PLANE_SLACK_ROWS;
    unsigned short* xh = ws.take<unsigned short>(planes_bytes(rx, I));   unsigned short* xl = xh + (size_t)rx * KpI;
    unsigned short* wih = ws.take<unsigned short>(planes_bytes(rw, I));  unsigned short* wil = wih + (size_t)rw * KpI;
    unsigned short* whh = ws.take<unsigned short>(planes_bytes(rw, H));  unsigned short* whl = whh + (size_t)rw * KpH;
    unsigned short* hp_[2]; unsigned short* hl_[2];
    for (int i = 0; i < 2; ++i) { hp_[i] = ws.take<unsigned short>(planes_bytes(rh, H)); hl_[i] = hp_[i] + (size_t)rh * KpH; }
    unsigned* sync_words = ws.take<unsigned>(GP_SYNC_BYTES);
    int rc = CTI_OK;
    if (!x16) { rc = split_planes(x, I, (int64_t)B * T, I, xh, xl, rx, st); if (rc) return rc; }
    if (w_ih_planes) { wih = static_cast<unsigned short*>(const_cast<void*>(w_ih_planes)); wil = wih + (size_t)rw * KpI; }   // resident planes
    else { rc = split_planes(w_ih, I, H3, I, wih, wil, rw, st); if (rc) return rc; }
    if (w_hh_planes) { whh = static_cast<unsigned short*>(const_cast<void*>(w_hh_planes)); whl = whh + (size_t)rw * KpH; }
    else { rc = split_planes(w_hh, H, H3, H, whh, whl, rw, st); if (rc) return rc; }
    if (KpH != H) {                                                       // K tail of the h planes stays zero (the kernel writes [0, H) only)
        for (int i = 0; i < 2; ++i) { rc = zero_planes(hp_[i], planes_bytes(rh, H), st); if (rc) return rc; }
    }
    {
        PlaneGemmArgs g{};
        g.Ah = xh; g.Al = xl; g.Bh = wih; g.Bl = wil; g.rows_allocA = rx; g.rows_allocB = rw; g.nb1 = 1; g.nb2 = 1;
        if (x16) {
            g.Abf = x16; g.ldabf = ldx16; g.rows_allocA = (int64_t)B * T;
            g.M = B * T; g.N = H3; g.Kp = KpI; g.terms = terms; g.epi = 0; g.C = gi; g.ldc_m = H3; g.ldc_n = 1; g.scale_div = 1; g.bias = b_ih;
            if (!gemm16_eligible(g)) {                              // (a shape cti_gemm16.hip refuses, e.g. K < 128 with a bias: the rows through planes after all)
                g.Abf = nullptr; g.ldabf = 0; g.rows_allocA = rx;
                rc = split_planes16(static_cast<const unsigned short*>(x16), ldx16, (int64_t)B * T, KpI, xh, nullptr, rx, st); if (rc) return rc;
            }
        }
        g.M = B * T; g.N = H3; g.Kp = KpI; g.terms = terms; g.epi = 0; g.C = gi; g.ldc_m = H3; g.ldc_n = 1; g.scale_div = 1; g.bias = b_ih;
        rc = gemm_nt_planes(g, st); if (rc) return rc;
    }
    if (gru_persistent_fits(B, H, prec, save)) {
        static thread_local int pattr_dev = -1;
        int dev = 0;
        (void)hipGetDevice(&dev);
        const size_t lds = (size_t)(KpH / 32) * 3072 + 4 * GP_TR;
        if (pattr_dev != dev) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 3072 + 4 * GP_TR);
            if (e != hipSuccess) return fail((int)e, "cti_gru_forward: hipFuncSetAttribute: %s", hipGetErrorString(e));
            pattr_dev = dev;
        }
        static const unsigned spin_limit = [] { const char* e = getenv("CTI_GRU_PERSISTENT_SPINS"); return e ? (unsigned)atol(e) : 400000u; }();
        rc = zero_planes(reinterpret_cast<unsigned short*>(sync_words), GP_SYNC_BYTES, st); if (rc) return rc;
        hipLaunchKernelGGL(gru_persistent_kernel, dim3(H / 16, B / 64), dim3(256), lds, st, hp_[0], hp_[1], rh * 16, (size_t)rh * KpH * 2, whh, rw * 16, KpH / 32,
                           gi, (int64_t)T * H3, b_hh, out, (int64_t)T * H, T, H, sync_words, spin_limit);
        return launch_status("cti_gru_forward/persistent");
    }
#ifndef CTI_GRU_FUSED
#define CTI_GRU_FUSED 1
#endif
    const bool fused = CTI_GRU_FUSED && KpH % 32 == 0 && (H + 15) / 16 <= 65535 && (B + 63) / 64 <= 65535;
    if (fused) {
        static thread_local int attr_dev = -1;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (attr_dev != dev) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_fused_kernel<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, GF_NST * GF_SLOT);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_fused_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, GfRing<1>::NST * GfRing<1>::SLOT);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_fused_kernel<3, GF_KPB>), hipFuncAttributeMaxDynamicSharedMemorySize, GF_KPB * GF_NST * GF_SLOT);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_step_fused_kernel<1, GF_KPB>), hipFuncAttributeMaxDynamicSharedMemorySize, GF_KPB * GfRing<1>::NST * GfRing<1>::SLOT);
            if (e != hipSuccess) return fail((int)e, "cti_gru_forward: hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_dev = dev;
        }
    }
    // per-step kernel: the K-split form (round 6) unless the caller asks for the ring form (CTI_TUNE_GRU_PERSISTENT = 2: the persistent form's bit-exact reference;
    // CTI_GRU_STEP_RING=1: A/B)
    static const bool ring_env = [] { const char* e = getenv("CTI_GRU_STEP_RING"); return e && e[0] == '1'; }();
    const bool ks_form = fused && !ring_env && tuning_gru_persistent() != 2;
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, prev = cur ^ 1;
        if (ks_form && t) {
            const dim3 grid((H + 15) / 16, (B + 63) / 64);
            if (terms == 3)
                hipLaunchKernelGGL(gru_step_ks_kernel<3>, grid, dim3(512), 0, st, hp_[prev], hl_[prev], rh * 16, whh, whl, rw * 16, KpH / 32, gi + (size_t)t * H3, (int64_t)T * H3, b_hh,
                                   h_tm + (size_t)(t - 1) * B * H, out + (size_t)t * H, (int64_t)T * H, h_tm + (size_t)t * B * H, save ? save + (size_t)t * B * 5 * H : nullptr,
                                   hp_[cur], hl_[cur], rh * 16, B, H);
            else
                hipLaunchKernelGGL(gru_step_ks_kernel<1>, grid, dim3(512), 0, st, hp_[prev], hl_[prev], rh * 16, whh, whl, rw * 16, KpH / 32, gi + (size_t)t * H3, (int64_t)T * H3, b_hh,
                                   h_tm + (size_t)(t - 1) * B * H, out + (size_t)t * H, (int64_t)T * H, h_tm + (size_t)t * B * H, save ? save + (size_t)t * B * 5 * H : nullptr,
                                   hp_[cur], hl_[cur], rh * 16, B, H);
            rc = launch_status("cti_gru_forward/K-split step"); if (rc) return rc;
            continue;
        }
        if (fused && t) {
            const dim3 grid((H + 15) / 16, (B + 63) / 64);
#define CTI_GF(TR) CTI_GF2(TR, 1)
#define CTI_GF2(TR, KP) hipLaunchKernelGGL((gru_step_fused_kernel<TR, KP>), grid, dim3(256), KP * GfRing<TR>::NST * GfRing<TR>::SLOT, st, hp_[prev], hl_[prev], rh * 16, whh, whl, rw * 16, KpH / 32, \
                                      gi + (size_t)t * H3, (int64_t)T * H3, b_hh, h_tm + (size_t)(t - 1) * B * H, out + (size_t)t * H, (int64_t)T * H,                     \
                                      h_tm + (size_t)t * B * H, save ? save + (size_t)t * B * 5 * H : nullptr, hp_[cur], hl_[cur], rh * 16, B, H)
            if ((KpH / 32) % GF_KPB == 0 && KpH / 32 >= 2 * GF_KPB) { if (terms == 3) CTI_GF2(3, GF_KPB); else CTI_GF2(1, GF_KPB); }
            else { if (terms == 3) CTI_GF(3); else CTI_GF(1); }
#undef CTI_GF
#undef CTI_GF2
            rc = launch_status("cti_gru_forward/fused step"); if (rc) return rc;
            continue;
        }
        if (t) { rc = step_gemm(hp_[prev], hl_[prev], rh, whh, whl, rw, B, H3, KpH, S, terms, part, st); if (rc) return rc; }
        hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(nblk), dim3(256), 0, st, gi + (size_t)t * H3, (int64_t)T * H3, part, t ? S : 0, b_hh,
                           t ? h_tm + (size_t)(t - 1) * B * H : nullptr, out + (size_t)t * H, (int64_t)T * H, h_tm + (size_t)t * B * H,
                           save ? save + (size_t)t * B * 5 * H : nullptr, hp_[cur], hl_[cur], rh * 16, B, H);
        rc = launch_status("cti_gru_forward/step"); if (rc) return rc;
    }
    return CTI_OK;
}

size_t cti_gru_backward_workspace_bytes(int B, int T, int H, int prec) {
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    const size_t f = sizeof(float);
    size_t n = 2 * al256(f * (size_t)B * H);                                                   // carry ping-pong
    if (prec == CTI_PREC_F32) return n + al256(f * (size_t)H * 3 * H) + al256(f * (size_t)B * H);
    const int S = plan_ksplit(B, H, planes_kp(3 * H), 1);
    n += al256(f * (size_t)S * B * H) + al256(f * (size_t)H * 3 * H);                          // partials, W_hh^T (fp32)
    n += al256(planes_bytes((int64_t)H + PLANE_SLACK_ROWS, 3 * H)) + al256(planes_bytes((int64_t)B + PLANE_SLACK_ROWS, 3 * H));
    return n;
}

int cti_gru_backward(const float* dout, const float* w_hh, const float* save, float* dgi, float* dgh, int B, int T, int H, int prec,
                     void* workspace, size_t workspace_bytes, void* stream) {
    CTI_REQUIRE_PTR(dout); CTI_REQUIRE_PTR(w_hh); CTI_REQUIRE_PTR(save); CTI_REQUIRE_PTR(dgi); CTI_REQUIRE_PTR(dgh); CTI_REQUIRE_PTR(workspace);
    CTI_REQUIRE(B > 0 && T > 0 && H > 0, CTI_E_SHAPE, "cti_gru_backward: B=%d T=%d H=%d", B, T, H);
    CTI_REQUIRE(prec == CTI_PREC_F32 || prec == CTI_PREC_BF16X3 || prec == CTI_PREC_BF16, CTI_E_UNSUPPORTED, "cti_gru_backward: prec=%d", prec);
    CTI_REQUIRE(workspace_bytes >= cti_gru_backward_workspace_bytes(B, T, H, prec), CTI_E_WORKSPACE, "cti_gru_backward: workspace too small");
    hipStream_t st = as_stream(stream);
    Carve ws{static_cast<char*>(workspace), 0};
    const size_t f = sizeof(float);
    const int H3 = 3 * H;
    float* carry[2] = {ws.take<float>(f * (size_t)B * H), ws.take<float>(f * (size_t)B * H)};
    const unsigned nblk = (unsigned)(((int64_t)B * H + 255) / 256);
    const bool planes = prec != CTI_PREC_F32;
    const int KpW = planes_kp(H3);
    const int S = planes ? plan_ksplit(B, H, KpW, 1) : 1;
    float* part = ws.take<float>(f * (size_t)S * B * H);
    float* wt = ws.take<float>(f * (size_t)H * H3);                         // W_hh^T (H, 3H): the contraction axis contiguous
    int rc = cti_transpose_f32(w_hh, H, 0, wt, H3, 0, H3, H, 1, stream); if (rc) return rc;
    unsigned short *wth = nullptr, *wtl = nullptr, *gh_ = nullptr, *gl_ = nullptr;
    const int64_t rw = (int64_t)H + PLANE_SLACK_ROWS, rg = (int64_t)B + PLANE_SLACK_ROWS;
    if (planes) {
        wth = ws.take<unsigned short>(planes_bytes(rw, H3)); wtl = wth + (size_t)rw * KpW;
        gh_ = ws.take<unsigned short>(planes_bytes(rg, H3)); gl_ = gh_ + (size_t)rg * KpW;
        rc = split_planes(wt, H3, H, H3, wth, wtl, rw, st); if (rc) return rc;
        if (KpW != H3) { rc = zero_planes(gh_, planes_bytes(rg, H3), st); if (rc) return rc; }
    }
    const int terms = prec == CTI_PREC_BF16X3 ? 3 : 1;
    for (int t = T - 1; t >= 0; --t) {
        const bool last = t == T - 1;
        float* dgh_t = dgh + (size_t)t * B * H3;                              // (T, B, 3H) time-major
        hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(nblk), dim3(256), 0, st, dout + (size_t)t * H, (int64_t)T * H, last ? nullptr : carry[(t + 1) & 1],
                           part, last ? 0 : S, save + (size_t)t * B * 5 * H, t ? save + (size_t)(t - 1) * B * 5 * H : nullptr,
                           dgi + (size_t)t * H3, (int64_t)T * H3, dgh_t, carry[t & 1], gh_, gl_, rg * 16, B, H);
        rc = launch_status("cti_gru_backward/step"); if (rc) return rc;
        if (t == 0) break;
        if (planes) {
            rc = step_gemm(gh_, gl_, rg, wth, wtl, rw, B, H, KpW, S, terms, part, st); if (rc) return rc;
        } else {
            GemmP q{};
            q.A = dgh_t; q.B = wt; q.C = part; q.lda = H3; q.ldb = H3; q.ldc_m = H; q.ldc_n = 1; q.nb1 = 1; q.nb2 = 1;
            q.M = B; q.N = H; q.K = H3; q.scale_div = 1;
            rc = gemm_nt_f32(q, st); if (rc) return rc;
        }
    }
    return CTI_OK;
}

}  // extern "C"
