// mb_gemm16: stand-alone prototype of the round-4 16-bit NT GEMM core (C[m][n] = sum_k A[m][k] B[n][k], bf16 in, fp32 accumulate).
//
// Structure (the "two wave groups, one interval apart" schedule of cdna_hip_programming.md section 5, rebuilt on 32-deep stages):
//   256 x 256 tile, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator registers);
//   an NS-slot LDS ring of 32-deep K stages (A 256 rows x 64 B | B 256 rows x 64 B = 32 KiB), filled by global_load_lds_dwordx4 (1 KiB per
//   wave-instruction = 16 rows x 64 B, source-side XOR swizzle so that the ds_read_b128 fragment reads are bank-conflict-free);
//   every wave alternates a LOAD interval (12 fragment reads of the stage, 4 DMA pieces of the stage NS-1 ahead, the previous tile's stores
//   when there are any, counted vmcnt + lgkmcnt(0)) with a COMPUTE interval (32 MFMAs), one raw s_barrier between intervals; waves 4-7 (the
//   SIMD partners of waves 0-3) run one interval behind, so on every SIMD one wave's MFMAs run beside the other's LDS reads and DMA issue.
//   A workgroup's tiles are ONE stream of stages: the ring never drains at a tile boundary and a tile's stores are issued in the LOAD
//   interval of the next tile's first stage, beside the partner group's MFMAs.
// build: hipcc -O3 --offload-arch=gfx950 -o mb_gemm16 mb_gemm16.hip        run: ./mb_gemm16 M N K [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef NS_RING
#define NS_RING 4
#endif
#ifndef MB_ABL           // timing-only ablations: 1 no DMA refill, 2 no MFMA, 4 no stores, 8 no LDS fragment reads (after the first)
#define MB_ABL 0
#endif
#ifndef MB_SETPRIO
#define MB_SETPRIO 1
#endif
#ifndef MB_STAGGER
#define MB_STAGGER 1
#endif
#ifndef MB_DMAC          // round 5 experiment: the wave's 4 LDS-DMA pieces of a stage are issued inside the COMPUTE interval, one per 8 MFMAs, instead of in the LOAD interval
#define MB_DMAC 0
#endif

struct P {
    const char* A; const char* B; float* C;
    int M, N, K;
    long long lda, ldb;          // row strides in BYTES
    long long ldc;               // row stride of C in elements
    int tiles_m, tiles_n, total_tiles, nk;
};

constexpr int BM = 256, BN = 256, BKB = 64;          // BKB: bytes of K per stage row (32 bf16)
constexpr int STAGE = (BM + BN) * BKB;               // 32 KiB
constexpr int NSTORE = 32;                            // epilogue stores per wave and tile (full tiles)

__device__ __forceinline__ void tile_coords(int id, int total, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    if (tiles_m <= tiles_n) { tm = vid % tiles_m; tn = vid / tiles_m; }
    else                    { tn = vid % tiles_n; tm = vid / tiles_n; }
}

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int V> __device__ __forceinline__ void wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(V) : "memory"); }
#define BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int NS>
__global__ __launch_bounds__(512) void gemm16_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    if ((int)blockIdx.x >= p.total_tiles) return;
    const int nk = p.nk;
    const int my_tiles = (p.total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int total = my_tiles * nk;

    // ---- fragment addresses inside a slot: lane (row lr of a 16-row MFMA tile, K group g of 8 elements = 16 B); chunk' = g ^ f(row), f = (-(row >> 2)) & 3
    const int lr = lane & 15, g = lane >> 4;
    const int fsw = (0 - (lr >> 2)) & 3;
    const int a_off = (wr * 128 + lr) * BKB + ((g ^ fsw) << 4);
    const int b_off = BM * BKB + (wc * 64 + lr) * BKB + ((g ^ fsw) << 4);

    // ---- DMA side: piece = 16 rows x 64 B; lane -> row (lane >> 2) of the piece, LDS chunk' = lane & 3 <- source chunk (lane & 3) ^ f(row)
    const int drow = lane >> 2, dch = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
    int iss_tile = blockIdx.x, iss_kb = 0, issued = 0;
    const char* Ab = nullptr; const char* Bb = nullptr;          // wave-uniform: the issue tile's operand origins at the current K stage
    unsigned voA0 = 0, voA1 = 0, voB0 = 0, voB1 = 0;             // per-lane byte offsets of the wave's four pieces
    auto issue_tile_setup = [&]() {
        int tm, tn;
        tile_coords(iss_tile, p.total_tiles, p.tiles_m, p.tiles_n, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        Ab = p.A + (long long)m0 * p.lda; Bb = p.B + (long long)n0 * p.ldb;
        const int ra0 = min(wid * 16 + drow, p.M - 1 - m0), ra1 = min(128 + wid * 16 + drow, p.M - 1 - m0);     // rows past the matrix re-read its last row
        const int rb0 = min(wid * 16 + drow, p.N - 1 - n0), rb1 = min(128 + wid * 16 + drow, p.N - 1 - n0);
        voA0 = (unsigned)(ra0 * p.lda) + dch * 16; voA1 = (unsigned)(ra1 * p.lda) + dch * 16;
        voB0 = (unsigned)(rb0 * p.ldb) + dch * 16; voB1 = (unsigned)(rb1 * p.ldb) + dch * 16;
#if MB_ABL & 16     // timing-only (WRONG data): every 1-KiB piece reads 8 rows x 128 B -- whole cache lines -- instead of 16 rows x 64 B: what line-granular DMA requests would buy
        {
            const int r8 = lane >> 3, c8 = lane & 7;
            voA0 = (unsigned)(min(wid * 16 + r8, p.M - 1 - m0) * p.lda) + c8 * 16; voA1 = (unsigned)(min(128 + wid * 16 + r8, p.M - 1 - m0) * p.lda) + c8 * 16;
            voB0 = (unsigned)(min(wid * 16 + r8, p.N - 1 - n0) * p.ldb) + c8 * 16; voB1 = (unsigned)(min(128 + wid * 16 + r8, p.N - 1 - n0) * p.ldb) + c8 * 16;
        }
#endif
    };
    auto issue_piece = [&](int slot, int q) {                    // piece q of the wave's four (A rows, A rows + 128, B rows, B rows + 128)
        if (issued >= total || (MB_ABL & 1)) return;
        char* sb = smem + slot * STAGE + wid * 1024;
        if (q == 0) glds16(Ab + voA0, sb);
        else if (q == 1) glds16(Ab + voA1, sb + 8192);
        else if (q == 2) glds16(Bb + voB0, sb + 16384);
        else glds16(Bb + voB1, sb + 24576);
    };
    auto issue_advance = [&]() {
        if (issued >= total || (MB_ABL & 1)) return;
        Ab += BKB; Bb += BKB;
        ++issued;
        if (++iss_kb == nk) {
            iss_kb = 0; iss_tile += (int)gridDim.x;
            if (iss_tile < p.total_tiles) issue_tile_setup();
        }
    };
    auto issue_next = [&](int slot) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_piece(slot, q);
        issue_advance();
    };

    issue_tile_setup();
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_next(s);
    if (total > NS - 2) wait_vm_lgkm0<4 * (NS - 2)>(); else wait_vm_lgkm0<0>();
    BARRIER();
#if MB_STAGGER
    if (wr == 1) BARRIER();                                      // waves 4-7 run one interval behind
#endif

    f32x4 acc[8][4];
    bf16x8 fa[8], fb[4];
    int vtile = blockIdx.x, kb = 0, slot = 0;
    int ep_tm = 0, ep_tn = 0, ep_age = 1000;                     // pending epilogue: tile coordinates; L intervals since its stores were issued
    bool ep_pending = false, ep_full = false;

    auto epilogue = [&](int tm, int tn, bool& full) {
        const int m0 = tm * BM + wr * 128, n0 = tn * BN + wc * 64;
        full = (tm * BM + BM <= p.M) && (tn * BN + BN <= p.N);
        // transposed product: lane holds C[m = 16 i + (lane & 15)][n = 16 j + 4 (lane >> 4) + 0..3]
        float* cp = p.C + (long long)(m0 + lr) * p.ldc + n0 + 4 * g;
        if (full) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!((MB_ABL & 4) && acc[i][j][0] != 12345.f)) *reinterpret_cast<f32x4*>(cp + j * 16) = acc[i][j];
                cp += 16 * p.ldc;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = m0 + i * 16 + lr;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + j * 16 + 4 * g;
                    if (m < p.M) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) cp[j * 16 + e] = acc[i][j][e];
                    }
                }
                cp += 16 * p.ldc;
            }
        }
    };

    for (int i = 0; i < total; ++i) {
        // ================= LOAD interval =================
        {
            const char* s = smem + slot * STAGE;
            if (!(MB_ABL & 8) || i == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) fa[u] = *reinterpret_cast<const bf16x8*>(s + a_off + u * 1024);
#pragma unroll
                for (int u = 0; u < 4; ++u) fb[u] = *reinterpret_cast<const bf16x8*>(s + b_off + u * 1024);
            }
        }
        const int rslot = slot == 0 ? NS - 1 : slot - 1;           // the slot stage i - 1 was read from: refilled with stage i + NS - 1
#if !MB_DMAC
        issue_next(rslot);
#elif MB_DMAC == 2          // two pieces here, two at the head of the COMPUTE interval (in front of its first MFMA: the matrix pipe is not running yet)
        issue_piece(rslot, 0); issue_piece(rslot, 1);
#endif
        if (ep_pending) {
            epilogue(ep_tm, ep_tn, ep_full);
            ep_pending = false; ep_age = 0;
        }
        {
            const int rem = total - 2 - i;                       // stages beyond i + 1 that exist; min(NS - 2, rem) of them may stay in flight
            const bool st = ep_full && ep_age <= NS - 2 - (MB_DMAC ? 1 : 0) && nk > NS - 2;      // (MB_DMAC: the stores are OLDER than the stage issued in the same iteration's COMPUTE interval: one interval less)
#if MB_DMAC == 2
            // in flight at this wait: stage i + 1, stage i + 2, the first two pieces of stage i + 3 (where they exist); i + 1 must have landed
            static_assert(NS == 4, "MB_DMAC: counted waits written for NS = 4");
            if (rem >= 2) { if (st) wait_vm_lgkm0<6 + NSTORE>(); else wait_vm_lgkm0<6>(); }
            else if (rem == 1) { if (st) wait_vm_lgkm0<4 + NSTORE>(); else wait_vm_lgkm0<4>(); }
            else { if (st) wait_vm_lgkm0<NSTORE>(); else wait_vm_lgkm0<0>(); }
#elif MB_DMAC
            // stage i + NS - 1 is issued in THIS iteration's COMPUTE interval: at this wait the stages i + 1 .. i + NS - 2 are in flight; i + 1 must have landed
            static_assert(!MB_DMAC || NS == 4, "MB_DMAC: counted waits written for NS = 4");
            if (rem >= 1) { if (st) wait_vm_lgkm0<4 + NSTORE>(); else wait_vm_lgkm0<4>(); }
            else { if (st) wait_vm_lgkm0<NSTORE>(); else wait_vm_lgkm0<0>(); }
#else
            if (NS >= 4 && rem >= 2) { if (st) wait_vm_lgkm0<8 + NSTORE>(); else wait_vm_lgkm0<(NS >= 4 ? 8 : 0)>(); }
            else if (NS >= 3 && rem == 1) { if (st) wait_vm_lgkm0<4 + NSTORE>(); else wait_vm_lgkm0<4>(); }
            else { if (st) wait_vm_lgkm0<NSTORE>(); else wait_vm_lgkm0<0>(); }
#endif
            ++ep_age;
        }
        BARRIER();
        // ================= COMPUTE interval =================
#if MB_DMAC >= 2
#if MB_DMAC == 3
        issue_piece(rslot, 0); issue_piece(rslot, 1);
#endif
        issue_piece(rslot, 2); issue_piece(rslot, 3);
        __builtin_amdgcn_sched_barrier(0);
#endif
#if MB_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#if MB_ABL & 2
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(fa[u]));
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" ::"v"(fb[u]));
        if (kb == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#else
        if (kb == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[v], fa[u], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[v], fa[u], acc[u][v], 0, 0, 0);
#if MB_DMAC == 1
                if (u & 1) { __builtin_amdgcn_sched_barrier(0); issue_piece(rslot, u >> 1); __builtin_amdgcn_sched_barrier(0); }     // one piece behind every 8 MFMAs
#endif
            }
        }
#endif
#if MB_DMAC == 1
        if (kb == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) issue_piece(rslot, q);      // (a tile's first stage takes the other MFMA branch above: its pieces here)
        }
#endif
#if MB_DMAC
        issue_advance();
#endif
#if MB_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        slot = slot == NS - 1 ? 0 : slot + 1;
        if (++kb == nk) {
            kb = 0;
            tile_coords(vtile, p.total_tiles, p.tiles_m, p.tiles_n, ep_tm, ep_tn);
            ep_pending = true;
            vtile += (int)gridDim.x;
        }
        BARRIER();
    }
    if (ep_pending) epilogue(ep_tm, ep_tn, ep_full);
#if MB_STAGGER
    if (wr == 0) BARRIER();
#endif
}

static inline uint16_t f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    if (K % 32) { printf("K must be a multiple of 32\n"); return 1; }
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : hA) x = f2bf(rnd());
    for (auto& x : hB) x = f2bf(rnd());
    char *dA, *dB; float* dC;
    CK(hipMalloc(&dA, hA.size() * 2 + 4096)); CK(hipMalloc(&dB, hB.size() * 2 + 4096)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
    P p{};
    p.A = dA; p.B = dB; p.C = dC; p.M = M; p.N = N; p.K = K; p.lda = (long long)K * 2; p.ldb = (long long)K * 2; p.ldc = N;
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN; p.total_tiles = p.tiles_m * p.tiles_n; p.nk = K / 32;
    auto kern = gemm16_kernel<NS_RING>;
    const int lds = NS_RING * STAGE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = p.total_tiles < ncu ? p.total_tiles : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    CK(hipDeviceSynchronize());
    // check: 2048 random entries + the four corners against a double-precision dot product of the bf16 inputs
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0; int bad = 0;
    auto check = [&](int m, int n) {
        double r = 0;
        for (int k = 0; k < K; ++k) r += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
        const double e = fabs(r - hC[(size_t)m * N + n]);
        if (!(e <= 1e-3 * sqrt((double)K))) { if (bad < 5) printf("  mismatch at (%d, %d): %g vs %g\n", m, n, hC[(size_t)m * N + n], r); ++bad; }
        if (e > maxerr) maxerr = e;
        if (fabs(r) > maxref) maxref = fabs(r);
    };
    check(0, 0); check(M - 1, N - 1); check(0, N - 1); check(M - 1, 0);
    for (int t = 0; t < 2048; ++t) { s = s * 1664525u + 1013904223u; const int m = (s >> 4) % M; s = s * 1664525u + 1013904223u; const int n = (s >> 4) % N; check(m, n); }
    // every row and column once (diagonal sweeps): catches a wrong tile / wave / lane mapping anywhere
    for (int m = 0; m < M; m += 1) check(m, (int)(((long long)m * 7919) % N));
    for (int n = 0; n < N; n += 1) check((int)(((long long)n * 104729) % M), n);
    printf("check: max |err| %.3e (max |ref| %.3e), %d mismatches\n", maxerr, maxref, bad);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("mb_gemm16 NS=%d ABL=%d prio=%d stag=%d  %d x %d x %d: %.1f us, %.1f TFLOP/s  (%d tiles on %d workgroups)\n", NS_RING, MB_ABL, MB_SETPRIO, MB_STAGGER, M, N, K, us,
           2.0 * M * N * K / us * 1e-6, p.total_tiles, grid);
    return bad != 0;
}
