#!/usr/bin/env python3
"""The GRU forward of the plain-bf16 mode: one launch per step (K-split step kernel = the default; LDS-ring step kernel of rounds 3-5) against the persistent form
(CTI_TUNE_GRU_PERSISTENT), same inputs, the persistent form compared bit for bit with the ring form;
time per call from torch events around back-to-back calls.

    python tools/bench_gru.py [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd  # noqa: E402
from cti_amd import ops  # noqa: E402
L = ops.L


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
    return e0.elapsed_time(e1) / reps * 1e3


def main(reps=20):
    torch.manual_seed(0)
    lib = L.lib()
    for (B, T, I, H) in ((256, 12, 600, 1024), (256, 6, 600, 1024), (256, 14, 600, 1024), (64, 5, 300, 512), (128, 3, 64, 1024), (256, 2, 32, 64)):
        x = torch.randn(B, T, I, device="cuda")
        k = 1.0 / H ** 0.5
        w_ih = (torch.rand(3 * H, I, device="cuda") * 2 - 1) * k; w_hh = (torch.rand(3 * H, H, device="cuda") * 2 - 1) * k
        b_ih = (torch.rand(3 * H, device="cuda") * 2 - 1) * k; b_hh = (torch.rand(3 * H, device="cuda") * 2 - 1) * k
        run = lambda: ops.gru_forward(x, w_ih, w_hh, b_ih, b_hh, prec="bf16")[0]
        L.check(lib.cti_set_tuning(L.TUNE_GRU_PERSISTENT, 0), "tuning")
        ks = run(); tk = timeit(run, reps)
        L.check(lib.cti_set_tuning(L.TUNE_GRU_PERSISTENT, 2), "tuning")
        ref = run(); t0 = timeit(run, reps)
        L.check(lib.cti_set_tuning(L.TUNE_GRU_PERSISTENT, 1), "tuning")
        got = run(); torch.cuda.synchronize()
        same = bool(torch.equal(ref, got)); nan = int(torch.isnan(got).sum())
        for _ in range(20):                       # again, under load from its own predecessors
            same = same and bool(torch.equal(ref, run()))
        t1 = timeit(run, reps)
        L.check(lib.cti_set_tuning(L.TUNE_GRU_PERSISTENT, 0), "tuning")
        print(json.dumps(dict(B=B, T=T, I=I, H=H, per_step_k_split_us=round(tk, 1), per_step_ring_us=round(t0, 1), persistent_us=round(t1, 1),
                              persistent_bit_identical_to_ring=same, nan=nan, k_split_vs_ring_max_abs_diff=float((ref - ks).abs().max()))), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
