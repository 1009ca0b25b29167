#!/usr/bin/env python3
"""HBM-roofline check of the path's memory-bound kernels (weighted sum-pools K5/K6, masked softmax K4, BiAttention
logits K7, zero-row mask) at B = 256: achieved GB/s = algorithmic bytes (inputs read once + outputs written once) / kernel time
(torch events around back-to-back launches), as a fraction of the 8 TB/s HBM3E peak.  One JSON line per kernel.

    python tools/bench_pools.py [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd  # noqa: E402
from cti_amd import ops  # noqa: E402

DEV = "cuda"
PEAK = 8000.0      # GB/s


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3       # us


def report(name, us, nbytes, **kw):
    gbs = nbytes / us / 1e3
    rec = dict(kernel=name, us=round(us, 2), MB=round(nbytes / 1e6, 2), GBps=round(gbs, 1), frac_of_hbm_peak=round(gbs / PEAK, 3), **kw)
    if nbytes < 256e6:
        #  a buffer this small is re-read out of the 256 MiB Infinity Cache by back-to-back launches: the rate is an algorithmic-byte rate,
        # NOT evidence of HBM traffic (rows_equal_prev on a 75 MB tensor reads "1.4 of the HBM peak" this way)
        rec["cache_resident"] = True
        rec["note"] = (rec.get("note", "") + "; " if rec.get("note") else "") + "operands fit the 256 MiB Infinity Cache: back-to-back launches re-read them on-die, frac_of_hbm_peak is not an HBM measurement"
    print(json.dumps(rec), flush=True)


def main(reps=20):
    torch.manual_seed(0)
    B, V, D = 256, 36, 1024
    f = 4
    for (Q, A) in ((14, 3), (12, 6)):
        vt = torch.randn(B, V, D, device=DEV); qt = torch.randn(B, Q, D, device=DEV); at = torch.randn(B, A, D, device=DEV)
        att = torch.softmax(torch.randn(B, V * Q * A, 2, device=DEV), 1).view(B, V, Q, A, 2)
        w = att[..., 0]
        report("tri_pool (TCNet.forward_with_weights pool)", timeit(lambda: ops.tri_pool(vt, qt, at, w), reps),
               f * B * (V * D + Q * D + A * D + V * Q * A + D), Q=Q, A=A, D=D)
    Q = 14
    vt = torch.randn(B, V, D, device=DEV); qt = torch.randn(B, Q, D, device=DEV)
    att = torch.softmax(torch.randn(B, 8, V * Q, device=DEV), 2).view(B, 8, V, Q)
    report("bi_pool k=1 (BCNet.forward_with_weights)", timeit(lambda: ops.bi_pool(vt, qt, att[:, 0], 1), reps), f * B * (V * D + Q * D + V * Q + D), Q=Q, D=D)
    D3 = 3072
    vt3 = torch.randn(B, V, D3, device=DEV); qt3 = torch.randn(B, Q, D3, device=DEV)
    report("bi_pool k=3", timeit(lambda: ops.bi_pool(vt3, qt3, att[:, 0], 3), reps), f * B * (V * D3 + Q * D3 + V * Q + D3 // 3), Q=Q, D=D3)
    h = torch.randn(8, D3, device=DEV); hb = torch.randn(8, device=DEV); hs = torch.ones(1, device=DEV)
    report("bi_logits G=8 (BiAttention logits)", timeit(lambda: ops.bi_logits(vt3, qt3, h, hs, hb), reps),
           f * (B * (V * D3 + Q * D3 + 8 * V * Q) + 8 * D3), Q=Q, D=D3, G=8)
    cti_amd.set_precision("bf16")
    report("bi_logits G=8, plain-bf16 mode (one product per pair)", timeit(lambda: ops.bi_logits(vt3, qt3, h, hs, hb), reps),
           f * (B * (V * D3 + Q * D3 + 8 * V * Q) + 8 * D3), Q=Q, D=D3, G=8)
    # round 5: the plain-bf16 mode's consumers reading the projected v as bf16 ROWS (what the hoisted projection GEMMs write): bytes = what each form really moves
    vt3h = vt3.to(torch.bfloat16)
    report("bi_logits G=8, plain-bf16 mode, bf16 vt rows (round 5)", timeit(lambda: ops.bi_logits(vt3h, qt3, h, hs, hb), reps),
           B * (2 * V * D3 + f * Q * D3 + f * 8 * V * Q) + f * 8 * D3, Q=Q, D=D3, G=8)
    vth = vt.to(torch.bfloat16)
    qadd = torch.randn(B, D, device=DEV) * 0.1
    report("bi_pool_shift (hoisted BAN loop), fp32 vt", timeit(lambda: ops.bi_pool_shift(vt, qt, qadd, att[:, 0]), reps), f * B * (V * D + Q * D + V * Q + 2 * D), Q=Q, D=D)
    report("bi_pool_shift, bf16 vt rows (round 5)", timeit(lambda: ops.bi_pool_shift(vth, qt, qadd, att[:, 0]), reps), B * (2 * V * D + f * (Q * D + V * Q + 2 * D)), Q=Q, D=D)
    slab = torch.randn(31, B, D, device=DEV) * 0.01
    adds = [(slab[i].data_ptr(), D) for i in range(31)]
    outm = torch.empty(B, 8 * D, device=DEV)
    report("bi_pool_shift_multi, bf16 vt rows + 31 addends (unrolled BAN loop, last glimpse)",
           timeit(lambda: ops.bi_pool_shift_multi(vth, qt, adds, att[:, 0], outm[:, :D]), reps), B * (2 * V * D + f * (Q * D + V * Q + 32 * D)), Q=Q, D=D)
    for (Qt, At) in ((14, 3), (12, 6)):
        qtt = torch.randn(B, Qt, D, device=DEV); att_ = torch.randn(B, At, D, device=DEV)
        wt = torch.softmax(torch.randn(B, V * Qt * At, device=DEV), 1).view(B, V, Qt, At)
        aadd = torch.randn(B, D, device=DEV) * 0.1
        report("tri_pool_shift (hoisted CTI loop), fp32 vt", timeit(lambda: ops.tri_pool_shift(vt, qtt, att_, qadd, aadd, wt), reps),
               f * B * (V * D + Qt * D + At * D + V * Qt * At + 3 * D), Q=Qt, A=At, D=D)
        report("tri_pool_shift, bf16 vt rows (round 5)", timeit(lambda: ops.tri_pool_shift(vth, qtt, att_, qadd, aadd, wt), reps),
               B * (2 * V * D + f * (Qt * D + At * D + V * Qt * At + 3 * D)), Q=Qt, A=At, D=D)
    cti_amd.set_precision("bf16x3")
    v = torch.randn(B, V, 2048, device=DEV).abs(); v[:, 30:] = 0
    vh = v.to(torch.bfloat16)
    report("zero_row_mask, bf16 rows (round 5)", timeit(lambda: ops.zero_row_mask(vh), reps), 2 * B * V * 2048 + B * V)
    report("rows_equal_prev (repeated-image detection, fp32 v)", timeit(lambda: ops.rows_equal_prev(v), reps), f * B * V * 2048 + B)
    report("zero_row_mask", timeit(lambda: ops.zero_row_mask(v), reps), f * B * V * 2048 + B * V)
    mask = ops.zero_row_mask(v)
    lg = torch.randn(B, 8, V, Q, device=DEV)
    report("masked_softmax_bi G=8", timeit(lambda: ops.masked_softmax_bi_(lg.clone(), mask), reps), 2 * f * B * 8 * V * Q, note="includes a clone of the logits")
    # round 3: mask + softmax in the logits kernel's last workgroup per sample (cti_biattention_fwd) -- built, parity-green, measured SLOWER than two launches; off by default
    import os
    os.environ["CTI_BIATT_FUSED"] = "0"
    t_sep = timeit(lambda: ops.biattention_forward(vt3, qt3, h, hs, hb, mask), reps)
    os.environ["CTI_BIATT_FUSED"] = "1"
    t_fus = timeit(lambda: ops.biattention_forward(vt3, qt3, h, hs, hb, mask), reps)
    os.environ.pop("CTI_BIATT_FUSED")
    report("BiAttention logits + mask + softmax, ONE launch (G=8)", t_fus, f * (B * (V * D3 + Q * D3 + 2 * 8 * V * Q) + 8 * D3), Q=Q, D=D3, G=8, separate_launches_us=round(t_sep, 2))
    for (Q, A, tag) in ((14, 3, "C4"), (14, 3129, "C2")):
        Bc = B if A < 100 else 64
        lg = torch.randn(Bc, V, Q, A, 2, device=DEV)
        m = ops.zero_row_mask(v[:Bc])
        buf = torch.empty_like(lg)

        def run():
            buf.copy_(lg)
            ops.masked_softmax_tri_(buf, m)
        t_all = timeit(run, reps)
        t_copy = timeit(lambda: buf.copy_(lg), reps)
        report("masked_softmax_tri %s" % tag, t_all - t_copy, 2 * f * lg.numel(), B=Bc, A=A, note="copy time subtracted; bytes = read logits + write p")
    # C2 again with the partial pass taken from the mode-3 GEMM's accumulators (cti_tcnet_forward_sm, precision f16f6): what the softmax
    # costs is the GEMM's extra time plus combine + the one normalise pass
    old = cti_amd.get_precision()
    cti_amd.set_precision("f16f6")
    try:
        Bc, Q, A = 64, 14, 3129
        tc = cti_amd.TCNet(2048, 1024, 300, 512, 1, 32, 2).to(DEV).eval()
        vv, qq, aa = v[:Bc].contiguous(), torch.randn(Bc, Q, 1024, device=DEV), torch.randn(Bc, A, 300, device=DEV)
        with torch.no_grad():
            t_plain = timeit(lambda: tc(vv, qq, aa, _want_mask=True), reps)
            t_sm = timeit(lambda: tc(vv, qq, aa, _want_mask=True, _want_sm_partials=True), reps)
            lg, m, part = tc(vv, qq, aa, _want_mask=True, _want_sm_partials=True)
            assert part is not None

            def core_ms(sm):                                          # the mode-3 GEMM alone: hipEvents the library records around its launch
                for _ in range(3):
                    tc(vv, qq, aa, _want_mask=True, _want_sm_partials=sm)
                ops.profile_start()
                for _ in range(reps):
                    tc(vv, qq, aa, _want_mask=True, _want_sm_partials=sm)
                torch.cuda.synchronize()
                ts = ops.profile_stop()["paralind_core"]
                return sorted(ts)[len(ts) // 2] * 1e3
            g_plain, g_sm = core_ms(False), core_ms(True)
            src = lg.clone()
            t_pass = timeit(lambda: (lg.copy_(src), ops.masked_softmax_tri_from_partials_(lg, m, part)), reps) - timeit(lambda: lg.copy_(src), reps)
        report("masked_softmax_tri C2, partial pass in the mode-3 epilogue", (g_sm - g_plain) + t_pass, 2 * f * lg.numel(), B=Bc, A=A,
               gemm_extra_us=round(g_sm - g_plain, 1), mode3_gemm_us=round(g_plain, 1), combine_normalise_us=round(t_pass, 1),
               forward_delta_us=round(t_sm - t_plain, 1), tcnet_forward_us=round(t_plain, 1),
               note="bytes = read logits + write p; time = (mode-3 GEMM with partials - without, medians of the library's hipEvents) + combine + normalise")
    finally:
        cti_amd.set_precision(old)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
