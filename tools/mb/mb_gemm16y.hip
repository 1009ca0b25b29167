// mb_gemm16y: ONE-wave-per-SIMD variant (see the kernel's comment) of mb_gemm16: stand-alone prototype of the round-4 16-bit NT GEMM core (C[m][n] = sum_k A[m][k] B[n][k], bf16 in, fp32 accumulate).
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

// ---- variant Y: ONE wave per SIMD.  4 waves of 128 x 128 (4 x 4 MFMA tiles of v_mfma_f32_32x32x16_bf16: 256 accumulator registers), two fragment sets; the
// LDS reads of stage s + 1 and the DMA pieces of stage s + 4 are interleaved into the MFMA stream of stage s, one barrier per stage in the middle of the stream.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NS>
__global__ __launch_bounds__(256) void gemm16_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    if ((int)blockIdx.x >= p.total_tiles) return;
    const int nk = p.nk;
    const int my_tiles = (p.total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int total = my_tiles * nk;
    // 32 x 32 x 16 operand: lane (row r = lane & 31, half h = lane >> 5) holds k = 8 h .. 8 h + 7 of K step ks: 16-B unit 2 ks + h of the row's 64 B,
    // stored at unit ^ ((row >> 2) & 3)
    const int r = lane & 31, h = lane >> 5;
    const int sw = (r >> 2) & 3;
    const int a_off0 = (wr * 128 + r) * BKB + (((0 + h) ^ sw) << 4), a_off1 = (wr * 128 + r) * BKB + (((2 + h) ^ sw) << 4);
    const int b_off0 = BM * BKB + (wc * 128 + r) * BKB + (((0 + h) ^ sw) << 4), b_off1 = BM * BKB + (wc * 128 + r) * BKB + (((2 + h) ^ sw) << 4);
    const int drow = lane >> 2, dch = (lane & 3) ^ ((lane >> 4) & 3);           // piece row (lane >> 2): (row >> 2) & 3 = (lane >> 4) & 3
    int iss_tile = blockIdx.x, iss_kb = 0, issued = 0;
    const char* Ab = nullptr; const char* Bb = nullptr;
    unsigned voA[4], voB[4];
    auto issue_tile_setup = [&]() {
        int tm, tn;
        tile_coords(iss_tile, p.total_tiles, p.tiles_m, p.tiles_n, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        Ab = p.A + (long long)m0 * p.lda; Bb = p.B + (long long)n0 * p.ldb;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            voA[u] = (unsigned)(min((wid + 4 * u) * 16 + drow, p.M - 1 - m0) * p.lda) + dch * 16;
            voB[u] = (unsigned)(min((wid + 4 * u) * 16 + drow, p.N - 1 - n0) * p.ldb) + dch * 16;
        }
    };
    auto issue_piece_nb = [&](int slot, int u) {
        if (MB_ABL & 1) return;
        char* sb = smem + slot * STAGE + (wid + 4 * (u & 3)) * 1024;
        if (u < 4) glds16(Ab + voA[u], sb); else glds16(Bb + voB[u & 3], sb + 16384);
    };
    auto issue_advance = [&]() {
        Ab += BKB; Bb += BKB;
        ++issued;
        if (++iss_kb == nk) {
            iss_kb = 0; iss_tile += (int)gridDim.x;
            if (iss_tile < p.total_tiles) issue_tile_setup();
        }
    };
    auto issue_piece = [&](int slot, int u) {
        if (issued >= total) return;
        issue_piece_nb(slot, u);
        if (u == 7) issue_advance();
    };
    issue_tile_setup();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int u = 0; u < 8; ++u) issue_piece(s, u);
    if (total >= NS) wait_vm_lgkm0<8 * (NS - 1)>(); else wait_vm_lgkm0<0>();
    BARRIER();

    f32x16 acc[4][4];
    bf16x8 fa[2][4][2], fb[2][4][2];                                 // [set][tile][K step]
    int vtile = blockIdx.x, kb = 0, slot = 0;
    auto read_frags = [&](int set, int sl) {
        const char* s = smem + sl * STAGE;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            fa[set][u][0] = *reinterpret_cast<const bf16x8*>(s + a_off0 + u * 2048); fa[set][u][1] = *reinterpret_cast<const bf16x8*>(s + a_off1 + u * 2048);
            fb[set][u][0] = *reinterpret_cast<const bf16x8*>(s + b_off0 + u * 2048); fb[set][u][1] = *reinterpret_cast<const bf16x8*>(s + b_off1 + u * 2048);
        }
    };
    auto epilogue = [&](int tm, int tn) {
        // transposed product: register e of tile (i, j) is C[m = 32 i + r][n = 32 j + 8 (e >> 2) + 4 h + (e & 3)]
        const int m0 = tm * BM + wr * 128, n0 = tn * BN + wc * 128;
        const bool full = (tm * BM + BM <= p.M) && (tn * BN + BN <= p.N);
        float* cp = p.C + (long long)(m0 + r) * p.ldc + n0 + 4 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + i * 32 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                f32x16 tv = acc[i][j];
                asm volatile("" : "+v"(tv));
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int n = n0 + j * 32 + 8 * e4 + 4 * h;
                    const f32x4 x = {tv[4 * e4], tv[4 * e4 + 1], tv[4 * e4 + 2], tv[4 * e4 + 3]};
                    if ((MB_ABL & 4) && x[0] != 12345.f) continue;
                    if (full) *reinterpret_cast<f32x4*>(cp + j * 32 + 8 * e4) = x;
                    else if (m < p.M) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) cp[j * 32 + 8 * e4 + e] = x[e];
                    }
                }
            }
            cp += 32 * p.ldc;
        }
    };
    read_frags(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

#define MF(set, u, v, ks) do { if (!(MB_ABL & 2)) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][v][ks], fa[set][u][ks], acc[u][v], 0, 0, 0); else asm volatile("" :: "v"(fb[set][v][ks]), "v"(fa[set][u][ks])); } while (0)
    // Body of stage i (fragments of stage i in set `set`): 8 MFMAs; wait for stage i + 1 (issued NS - 1 bodies ago) and meet the other three waves;
    // then the other 24 MFMAs with the 16 fragment reads of stage i + 1 and the 8 DMA pieces of stage i + NS (into the slot stage i just left) between them.
    // The wait: stages i + 2 .. i + NS - 1 may stay in flight (8 pieces each); in the first NS - 1 bodies after a full tile's epilogue its 64 stores are
    // younger than the stage needed and the window is clamped to 63.
#define STAGE_BODY(set, DMA)                                                                                                    \
    {                                                                                                                           \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) { MF(set, 0, v, 0); MF(set, 0, v, 1); }                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        if (DMA) {                                                                                                              \
            if (young_stores > 0) { asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); --young_stores; }                         \
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NS - 2)) : "memory");                                            \
        } else {                                                                                                                \
            const int rem = total - 2 - i;                                                                                      \
            if (rem >= 2 && NS == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                          \
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                 \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
        }                                                                                                                       \
        BARRIER();                                                                                                              \
        const int nslot = slot == NS - 1 ? 0 : slot + 1;                                                                        \
        if (!(MB_ABL & 8)) read_frags(1 - (set), nslot);                                                                        \
        _Pragma("unroll") for (int u = 1; u < 4; ++u) {                                                                         \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) { MF(set, u, v, 0); MF(set, u, v, 1); }                                \
        }                                                                                                                       \
        if (DMA) { _Pragma("unroll") for (int u = 0; u < 8; ++u) issue_piece_nb(slot, u); }                                     \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }             \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                      \
        if (DMA) issue_advance();                                                                                               \
        slot = nslot;                                                                                                           \
    }
#define ZERO_ACC()                                                                                                              \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) _Pragma("unroll") for (int v = 0; v < 4; ++v)                                  \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[u][v][e] = 0.f;
#define TILE_END()                                                                                                              \
    {                                                                                                                           \
        kb = 0;                                                                                                                 \
        int tm, tn;                                                                                                             \
        tile_coords(vtile, p.total_tiles, p.tiles_m, p.tiles_n, tm, tn);                                                        \
        const bool full = (tm * BM + BM <= p.M) && (tn * BN + BN <= p.N);                                                       \
        if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    /* ragged tile: its store count is not a constant */      \
        epilogue(tm, tn);                                                                                                       \
        if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
        young_stores = full && !(MB_ABL & 4) ? NS - 1 : 0;                                                                      \
        vtile += (int)gridDim.x;                                                                                                \
    }
    // nk even (K a multiple of 64): the two fragment sets alternate statically, a tile always ends after the second body.  The last NS bodies of the
    // workgroup issue nothing (their own loop, so the steady-state body has no branch around its DMA).
    int young_stores = 0;
    int i = 0;
    for (; i < total - NS; ++i) {
        if (kb == 0) { ZERO_ACC() }
        STAGE_BODY(0, 1)
        ++i;
        __builtin_amdgcn_sched_barrier(0);
        STAGE_BODY(1, 1)
        kb += 2;
        if (kb == nk) TILE_END()
    }
    young_stores = 0;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NS - 2)) : "memory");       // a tile ended right before: its stores are not to be counted in the tail's windows
    for (; i < total; ++i) {
        if (kb == 0) { ZERO_ACC() }
        STAGE_BODY(0, 0)
        ++i;
        __builtin_amdgcn_sched_barrier(0);
        STAGE_BODY(1, 0)
        kb += 2;
        if (kb == nk) TILE_END()
    }
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
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
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("mb_gemm16y NS=%d ABL=%d prio=%d stag=%d  %d x %d x %d: %.1f us, %.1f TFLOP/s  (%d tiles on %d workgroups)\n", NS_RING, MB_ABL, MB_SETPRIO, MB_STAGGER, M, N, K, us,
           2.0 * M * N * K / us * 1e-6, p.total_tiles, grid);
    return bad != 0;
}
