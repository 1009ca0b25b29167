#!/usr/bin/env python3
"""64-bit indexing check: TriAttention at the configs[1] widths with B = 700 (2.2e9 output elements, past 2^31) against the same module run
on 2-sample slices taken from the start, the middle and the end of the batch."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import cti_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 700
c = dict(bench.C2, B=B)
torch.manual_seed(1204)
att = cti_amd.TriAttention(c["v_dim"], c["q_dim"], c["a_dim"], c["h_mm"], 1, c["rank"], c["glimpse"], 1).cuda().eval()
v, q, a = bench.synth_inputs(c, B, 11, torch.device("cuda"))
with torch.no_grad():
    p, logits = att(v, q, a)
    print("elements", p.numel(), ">", 2 ** 31, "finite:", bool(torch.isfinite(p).all()))
    worst = 0.0
    for lo in (0, B // 2, B - 2):
        ps, ls = att(v[lo:lo + 2].contiguous(), q[lo:lo + 2].contiguous(), a[lo:lo + 2].contiguous())
        dp = float((p[lo:lo + 2] - ps).abs().max() / ps.abs().max())
        fin = torch.isfinite(ls)
        dl = float((logits[lo:lo + 2][fin] - ls[fin]).abs().max() / ls[fin].abs().max())
        same_inf = bool((torch.isfinite(logits[lo:lo + 2]) == fin).all())
        worst = max(worst, dp, dl)
        print("samples %d..%d: p %.2e logits %.2e mask pattern identical: %s" % (lo, lo + 1, dp, dl, same_inf))
    print("OK" if worst < 1e-5 else "MISMATCH", worst)
