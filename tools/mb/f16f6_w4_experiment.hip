// NOT BUILT -- record of a rejected experiment (round 2): the 4-wave form of gemm_f16f6_kernel (one wave per SIMD, 128 x 96 register tile,
// accumulators in AGPRs, operands reloaded in place as they die).  Correct (tests/test_f16f6_gpu.py passed), 2.27-2.36 ms against 2.00 ms for the 8-wave
// kernel at BASELINE configs[1]: with one wave per SIMD nothing fills the stalls of the non-MFMA instruction stream, and hipcc's register
// allocator parks operand tuples in the AGPR half / accumulator tiles in VGPRs (hundreds of v_accvgpr moves, scratch spills whose reloads
// drain vmcnt = the DMA ring) unless every class is pinned.  This text belongs inside cti_gemm_f16f6.hip's anonymous namespace.
// ---- 4-wave form: ONE wave per SIMD with a 128 x 96 register tile (192 accumulator registers in the AGPR half of the file).  The 8-wave
// kernel above is instruction-ISSUE bound -- a wave issues at most one instruction per four cycles, and its ~400 instructions per K block
// (DMA, addressing, fragment reads, conversions, 18 MFMAs) take longer to issue than the MFMAs take to run.  Here a wave has twice the
// MFMAs per K block (36) for 1.4x the fragment reads and the other instructions issue in the shadow of its own MFMAs: the fragments of
// block b + 1 are read (and their hi codes derived) while the MFMAs of block b run, ping-pong between two register sets.
// Ring protocol per block b:  [vmcnt: my pieces of b + 1 landed] [lgkmcnt(0): my reads of b done]  barrier  -> DMA of block b + 4 into
// b's slot -> LDS reads of b + 1 -> MFMAs of b (+ conversions of b + 1).  Four 40 KiB slots: one being read, three in flight.
template <int EPI, class G>
__global__ __launch_bounds__(G::NTHR) __attribute__((amdgpu_num_vgpr(448))) void gemm_f16f6_w4_kernel(F6P p) {
    constexpr int WN = G::WN, TM = G::TM, TN = G::TN, NST = G::NST, BM = G::BM, BN = G::BN, NW = G::NW, SLOT = G::SLOT, CNT = G::CNT;
    static_assert(NW == 4 && CNT == 10 && G::PAH == 16 && G::PBH == 12 && G::PAF == 6 && G::PBF == 4, "the piece schedule below is written out for 4 waves on 256 x 192");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    if ((int)blockIdx.x >= p.total_tiles) return;
    const int nkb = p.Kb;
    const int nblk = ((p.total_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1) * nkb;       // the workgroup's tiles are one stream of K blocks

    // ---- DMA side: piece g = wid + 4 u (u = 0 .. 9) of [A_H x16 | B_H x12 | A_FL x6 | B_FL x4 | B_FL tail + A_S | B_S].  u = 0-3 are A_H
    // pieces, 4-6 B_H, 7 A_FL for every wave; u = 8, 9 depend on the wave (scalar cursors c8, c9).  Cursors point at K block iss_kb.
    const char *cAH = nullptr, *cBH = nullptr, *cAF = nullptr, *c8 = nullptr, *c9 = nullptr, *cAS = nullptr;
    const int64_t ks8 = (wid < 2 ? p.pA : p.pB) * 24, ks9 = wid == 3 ? p.pBS * 2 : p.pB * 24;
    const int lds8 = wid < 2 ? G::OFF_AFL + (wid + 4) * 1024 : G::OFF_BFL + (wid - 2) * 1024;
    const int lds9 = wid == 3 ? G::OFF_BS : G::OFF_BFL + (wid + 2) * 1024;                        // wid == 2: the B_FL tail + the A scales
    int iss_tile = blockIdx.x, iss_kb = 0, issued = 0;
    auto issue_tile_setup = [&]() {
        int zz, tm, tn;
        tile_coords(iss_tile, p.total_tiles, tiles_m, tiles_n, zz, tm, tn);
        const int64_t ra = (int64_t)zz * p.rA + tm * BM, rb = (int64_t)zz * p.rB + tn * BN;
        cAH = p.AH + (ra + wid * 16) * 64; cBH = p.BH + (rb + wid * 16) * 64; cAF = p.AFL + ra * 24 + wid * 1024;
        c8 = wid < 2 ? p.AFL + ra * 24 + (wid + 4) * 1024 : p.BFL + rb * 24 + (wid - 2) * 1024;
        c9 = wid == 3 ? p.BS + rb * 2 : p.BFL + rb * 24 + (wid + 2) * 1024;
        cAS = p.AS + ra * 2 - 512;                                  // upper 32 lanes of the shared piece: 16 B per lane from lane 32 on
    };
    auto issue_pieces = [&](int pos) {                              // the DMA of the stream's next K block into ring slot `pos`
        if (issued >= nblk || (CTI_F6_ABL & 1)) return;
        char* slot = smem + pos * SLOT;
        // the per-lane source offsets are recomputed here (opaque to the optimiser): kept live across the block they would be the first
        // values to be spilled, and a scratch reload makes the compiler drain vmcnt -- the whole DMA ring -- in front of it
        unsigned l_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
        const unsigned hoff = (l_ >> 2) * 64 + (((l_ & 3) ^ ((l_ >> 4) & 3)) << 4), loff = l_ * 16u;
        const int h = l_ >> 5;
#pragma unroll
        for (int u = 0; u < 4; ++u) dma16(cAH + u * 4096 + hoff, slot + G::OFF_AH + (wid + 4 * u) * 1024);
#pragma unroll
        for (int u = 0; u < 3; ++u) dma16(cBH + u * 4096 + hoff, slot + G::OFF_BH + (wid + 4 * u) * 1024);
        dma16(cAF + loff, slot + G::OFF_AFL + wid * 1024);
        dma16(c8 + loff, slot + lds8);
        const char* s9 = c9;
        if (wid == 2) s9 = h ? cAS : c9;
        dma16(s9 + loff, slot + lds9);
    };
    auto issue_advance = [&]() {                                    // ... and the cursors' step to the block after it
        if (issued >= nblk) return;
        cAH += p.pA * 64; cBH += p.pB * 64; cAF += p.pA * 24; c8 += ks8; c9 += ks9; cAS += p.pAS * 2;
        ++issued;
        if (++iss_kb == nkb) {
            iss_kb = 0; iss_tile += (int)gridDim.x;
            if (iss_tile < p.total_tiles) issue_tile_setup();
        }
    };
    auto issue_next = [&](int pos) { issue_pieces(pos); issue_advance(); };
    // vmcnt part of "block b has landed" when the stream has been issued up to block b + ahead (or to its end): that many blocks' pieces may
    // still be in flight.  vmcnt retires in issue order, stores included: epilogue stores issued in between only make the wait conservative.
    auto wait_block = [&](int b, int ahead) {
        const int rem = nblk - 1 - b, n = rem < ahead ? rem : ahead;
        if (n >= 3) wait_vm<3 * CNT>(); else if (n == 2) wait_vm<2 * CNT>(); else if (n == 1) wait_vm<CNT>(); else wait_vm<0>();
    };

    if (p.desync_ticks > 0) {                                       // de-synchronise the CUs' epilogues (see the 8-wave kernel)
        const unsigned target = (((unsigned)blockIdx.x * 157u) & 255u) * (unsigned)p.desync_ticks >> 8;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((unsigned)(__builtin_amdgcn_s_memrealtime() - t0) < target) __builtin_amdgcn_s_sleep(16);
    }
#if CTI_F6_ABL & 128
    const unsigned long long probe_c0 = __builtin_readcyclecounter(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    issue_tile_setup();
#pragma unroll
    for (int i = 0; i < NST; ++i) issue_next(i);

    // ---- compute side
    int vtile = blockIdx.x, z = 0, m0 = 0, n0 = 0;
    auto compute_tile_setup = [&]() {
        int tm, tn;
        tile_coords(vtile, p.total_tiles, tiles_m, tiles_n, z, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
    };
    compute_tile_setup();
    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();

    // per-lane fragment addresses inside a slot; tile i / j of the wave adds a compile-time constant (32 rows), folded into the ds_read offsets
    const int sw = (r >> 2) & 3;
    const int c0 = ((0 + h) ^ sw) << 4, c1 = ((2 + h) ^ sw) << 4;
    const int rowA = wm * TM * 32 + r, rowB = wn * TN * 32 + r;
    const int aH0 = G::OFF_AH + rowA * 64 + c0, aH1 = G::OFF_AH + rowA * 64 + c1, bH0 = G::OFF_BH + rowB * 64 + c0, bH1 = G::OFF_BH + rowB * 64 + c1;
    const int aF = G::OFF_AFL + rowA * 24 + 12 * h, bF = G::OFF_BFL + rowB * 24 + 12 * h;
    const int aS = G::OFF_AS + ((rowA * 2) & ~3), bS = G::OFF_BS + ((rowB * 2) & ~3);
    const int shS = (r & 1) * 16;                                   // the row's (hi, lo) scale bytes inside that dword
    const int shA = shS + h * 8, shB = shS + (1 - h) * 8;           // byte the MFMA takes: A lower lanes hi / upper lo, B the other way round

    struct Frags { f16x8 a16[TM][2], b16[TN][2]; u32x6 a6[TM], b6[TN]; int sa[TM], sb[TN]; };
    struct Raw { int fa[TM][3], fb[TN][3], spa[TM], spb[TN]; };
#if CTI_F6_ABL & 64
#define CTI_F6_SWAP(x, y) u32x2{(unsigned)(x), (unsigned)(y)}
#else
#define CTI_F6_SWAP(x, y) __builtin_amdgcn_permlane32_swap((unsigned)(x), (unsigned)(y), false, false)
#endif
    // LDS reads of one A tile / of the B tiles of a block; fp6 operands: hi codes from the f16 fragment (one conversion), traded with the
    // SIMD-half partner against the lo codes read from LDS -- see the 8-wave kernel's fragment notes
    auto read_A = [&](Frags& f, Raw& w, const char* s, int i) {
        f.a16[i][0] = *reinterpret_cast<const f16x8*>(s + aH0 + i * 2048);
        f.a16[i][1] = *reinterpret_cast<const f16x8*>(s + aH1 + i * 2048);
#pragma unroll
        for (int d = 0; d < 3; ++d) w.fa[i][d] = *reinterpret_cast<const int*>(s + aF + i * 768 + 4 * d);
        w.spa[i] = *reinterpret_cast<const int*>(s + aS + i * 64);
    };
    auto read_B1 = [&](Frags& f, Raw& w, const char* s, int j) {
        f.b16[j][0] = *reinterpret_cast<const f16x8*>(s + bH0 + j * 2048);
        f.b16[j][1] = *reinterpret_cast<const f16x8*>(s + bH1 + j * 2048);
#pragma unroll
        for (int d = 0; d < 3; ++d) w.fb[j][d] = *reinterpret_cast<const int*>(s + bF + j * 768 + 4 * d);
        w.spb[j] = *reinterpret_cast<const int*>(s + bS + j * 64);
    };
    auto read_B = [&](Frags& f, Raw& w, const char* s) {
#pragma unroll
        for (int j = 0; j < TN; ++j) read_B1(f, w, s, j);
    };
    // (the conversion takes its own, transient copy of the f16 values: the instruction wants them at the bottom of a 16-register tuple,
    // and tying the long-lived MFMA operands into such tuples costs more registers than the second LDS read costs time)
    auto convert_A = [&](Frags& f, const Raw& w, const char* s, int i) {
        const f16x8 k0 = *reinterpret_cast<const f16x8*>(s + aH0 + i * 2048), k1 = *reinterpret_cast<const f16x8*>(s + aH1 + i * 2048);
        const u32x6 own = f6_codes_of_f16(k0, k1, w.spa[i] >> shS);
        const auto w0 = CTI_F6_SWAP(own[0], w.fa[i][0]); const auto w1 = CTI_F6_SWAP(own[1], w.fa[i][1]); const auto w2 = CTI_F6_SWAP(own[2], w.fa[i][2]);
        f.a6[i][0] = w0[0]; f.a6[i][1] = w1[0]; f.a6[i][2] = w2[0]; f.a6[i][3] = w0[1]; f.a6[i][4] = w1[1]; f.a6[i][5] = w2[1];
        f.sa[i] = w.spa[i] >> shA;
    };
    auto convert_B1 = [&](Frags& f, const Raw& w, const char* s, int j) {
        const f16x8 k0 = *reinterpret_cast<const f16x8*>(s + bH0 + j * 2048), k1 = *reinterpret_cast<const f16x8*>(s + bH1 + j * 2048);
        const u32x6 own = f6_codes_of_f16(k0, k1, w.spb[j] >> shS);
        const auto w0 = CTI_F6_SWAP(w.fb[j][0], own[0]); const auto w1 = CTI_F6_SWAP(w.fb[j][1], own[1]); const auto w2 = CTI_F6_SWAP(w.fb[j][2], own[2]);
        f.b6[j][0] = w0[0]; f.b6[j][1] = w1[0]; f.b6[j][2] = w2[0]; f.b6[j][3] = w0[1]; f.b6[j][4] = w1[1]; f.b6[j][5] = w2[1];
        f.sb[j] = w.spb[j] >> shB;
    };
    // The MFMAs are written as inline assembly so that the register classes are fixed: accumulators in the AGPR half of the register file,
    // operands in VGPRs.  (Through the builtins the allocator moved accumulator tiles into VGPRs and operand tuples into the AGPRs the
    // accumulators fill, and paid for it with hundreds of v_accvgpr moves and with scratch spills -- whose reloads drain vmcnt, i.e. the
    // DMA ring.)  The compiler does not see MFMA hazards inside the assembly: consecutive MFMAs on one accumulator use identical
    // vDst / SrcC (back-to-back accumulation is interlocked in hardware), operands come from LDS loads / VALU results (interlocked), and
    // the one software-managed case -- VALU reads of a freshly written accumulator -- is covered by the wait states in tile_end.
    auto mfma_f16 = [&](const Frags& f, int i, int j, int k) {
        if (!(CTI_F6_ABL & 2) || p.Kb < 0) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(f.a16[i][k]), "v"(f.b16[j][k]));
    };
    auto mfma_f6 = [&](const Frags& f, int i, int j) {
        if (!(CTI_F6_ABL & 2) || p.Kb < 0)
            asm("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
                : "+a"(acc[i][j]) : "v"(f.a6[i]), "v"(f.b6[j]), "v"(f.sa[i]), "v"(f.sb[j]));
    };
    auto mfma_drain = [&]() {
        // 19 wait states between a 16-pass MFMA's issue and a VALU read of its result (the accumulators are operands so that the
        // compiler keeps this behind the last MFMA of every tile)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7"
                     : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]),
                       "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]));
    };
    auto tile_end = [&](int b, bool more, bool& waited) {           // after the last block of a tile
        if (b + 2 < nblk) { wait_block(b + 2, NST - 2); waited = true; }           // the next barrier's wait, ahead of the stores
        mfma_drain();
        f6_epilogue<EPI, G>(acc, p, z, m0, n0, wm, wn, lane);
        if (more) { vtile += (int)gridDim.x; compute_tile_setup(); zero_acc(); }
    };

    // ---- the stream.  ONE operand set, reloaded in place as it dies (a second one does not fit beside 192 accumulators): the MFMAs of a
    // block go row by row; the wave's A tile i is dead once row i has issued and block b + 1's A tile i is read behind it (and converted
    // behind the next row); in the last row B tile j dies with tile (TM-1, j) and is re-read behind it.  The conversions of the last-read
    // tiles are the head of the next block, among row 0, whose f16 MFMAs need no conversion.
    // A single wave only overlaps its own MFMAs with its other instructions if they ALTERNATE in the instruction stream (an MFMA issues
    // when the matrix pipe accepts it; what follows issues in its shadow): the block is written as chunks of three independent MFMAs plus
    // a share of the other work, each closed by a scheduling barrier (CTI_F6_CHUNK) so that the compiler keeps the interleave.
    // The block's barrier comes BEHIND row 0: by then the wave's reads of block b are long complete (its slot may be refilled) and block
    // b + 1's pieces, issued three blocks earlier, have landed; the matrix pipe works through row 0's fp6 MFMAs while the waves meet.
#define CTI_F6_CHUNK() __builtin_amdgcn_sched_barrier(0)
    static_assert(TM == 4 && TN == 3 && (NST & (NST - 1)) == 0, "the chunk schedule is written out for a 4 x 3 register tile");
    int kb = 0;
    bool waited = false;
    Frags F; Raw W;
    wait_block(0, NST - 1);
    __builtin_amdgcn_s_barrier();
    read_B(F, W, smem);
#pragma unroll
    for (int i = 0; i < TM; ++i) read_A(F, W, smem, i);
#pragma unroll
    for (int i = 0; i < TM - 1; ++i) convert_A(F, W, smem, i);
    int b = 0;
    for (; b + 1 < nblk; ++b) {                                      // every block but the stream's last: block b + 1 exists
        const char* sc = smem + (b & (NST - 1)) * SLOT;
        const char* sn = smem + ((b + 1) & (NST - 1)) * SLOT;
        // row 0 and the conversions the previous block left
        mfma_f16(F, 0, 0, 0); mfma_f16(F, 0, 1, 0); mfma_f16(F, 0, 2, 0); convert_B1(F, W, sc, 0); CTI_F6_CHUNK();
        mfma_f16(F, 0, 0, 1); mfma_f16(F, 0, 1, 1); mfma_f16(F, 0, 2, 1); convert_B1(F, W, sc, 1); convert_B1(F, W, sc, 2); CTI_F6_CHUNK();
        mfma_f6(F, 0, 0); mfma_f6(F, 0, 1); mfma_f6(F, 0, 2); convert_A(F, W, sc, TM - 1); CTI_F6_CHUNK();
        if (!waited) wait_block(b + 1, NST - 2);
        waited = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of block b are complete
        __builtin_amdgcn_s_barrier();
        issue_pieces(b & (NST - 1));
        read_A(F, W, sn, 0); CTI_F6_CHUNK();
        // rows 1, 2: three independent MFMAs per chunk
        mfma_f6(F, 1, 0); mfma_f6(F, 1, 1); mfma_f6(F, 1, 2); CTI_F6_CHUNK();
        mfma_f16(F, 1, 0, 0); mfma_f16(F, 1, 1, 0); mfma_f16(F, 1, 2, 0); convert_A(F, W, sn, 0); CTI_F6_CHUNK();
        mfma_f16(F, 1, 0, 1); mfma_f16(F, 1, 1, 1); mfma_f16(F, 1, 2, 1); CTI_F6_CHUNK();
        read_A(F, W, sn, 1);
        mfma_f6(F, 2, 0); mfma_f6(F, 2, 1); mfma_f6(F, 2, 2); CTI_F6_CHUNK();
        mfma_f16(F, 2, 0, 0); mfma_f16(F, 2, 1, 0); mfma_f16(F, 2, 2, 0); convert_A(F, W, sn, 1); CTI_F6_CHUNK();
        mfma_f16(F, 2, 0, 1); mfma_f16(F, 2, 1, 1); mfma_f16(F, 2, 2, 1); CTI_F6_CHUNK();
        read_A(F, W, sn, 2);
        // row 3 tile by tile: B tile j dies with tile (3, j)
        mfma_f6(F, 3, 0); mfma_f16(F, 3, 0, 0); mfma_f16(F, 3, 0, 1); CTI_F6_CHUNK();
        read_B1(F, W, sn, 0);
        mfma_f6(F, 3, 1); mfma_f16(F, 3, 1, 0); mfma_f16(F, 3, 1, 1); convert_A(F, W, sn, 2); CTI_F6_CHUNK();
        read_B1(F, W, sn, 1);
        mfma_f6(F, 3, 2); mfma_f16(F, 3, 2, 0); mfma_f16(F, 3, 2, 1); CTI_F6_CHUNK();
        read_B1(F, W, sn, 2); read_A(F, W, sn, TM - 1);
        issue_advance();
        if (++kb == nkb) { kb = 0; tile_end(b, true, waited); }
    }
    {                                                               // the stream's last block: nothing left to read or to issue
        const char* sc = smem + (b & (NST - 1)) * SLOT;
#pragma unroll
        for (int j = 0; j < TN; ++j) { mfma_f16(F, 0, j, 0); mfma_f16(F, 0, j, 1); }
#pragma unroll
        for (int j = 0; j < TN; ++j) convert_B1(F, W, sc, j);
        convert_A(F, W, sc, TM - 1);
#pragma unroll
        for (int j = 0; j < TN; ++j) mfma_f6(F, 0, j);
#pragma unroll
        for (int i = 1; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) { mfma_f6(F, i, j); mfma_f16(F, i, j, 0); mfma_f16(F, i, j, 1); }
        tile_end(b, false, waited);
    }
#undef CTI_F6_CHUNK
#undef CTI_F6_SWAP
#if CTI_F6_ABL & 128
    if (t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.C[2 * blockIdx.x] = (float)(__builtin_readcyclecounter() - probe_c0); p.C[2 * blockIdx.x + 1] = (float)(__builtin_amdgcn_s_memrealtime() - probe_r0);
    }
#endif
}

