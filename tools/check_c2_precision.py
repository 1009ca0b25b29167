#!/usr/bin/env python3
"""One-off numerics check at the BASELINE configs[1] widths (A = 3129, B = 4): the fp32-grade split-bf16 mode and the plain-bf16 mode against
the exact-fp32 MFMA mode of the same library (normalised max error of TCNet.forward and of the TriAttention map)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import cti_amd  # noqa: E402

c = dict(bench.C2, B=4)
torch.manual_seed(1204)
att = cti_amd.TriAttention(c["v_dim"], c["q_dim"], c["a_dim"], c["h_mm"], 1, c["rank"], c["glimpse"], 1).cuda().eval()
v, q, a = bench.synth_inputs(c, c["B"], 7, torch.device("cuda"))
res = {}
for mode in ("fp32", "bf16x3", "bf16"):
    cti_amd.set_precision(mode)
    with torch.no_grad():
        raw = att.TriAtt(v, q, a)
        p, _ = att(v, q, a)
    res[mode] = (raw.double(), p.double())
for mode in ("bf16x3", "bf16"):
    e_raw = float((res[mode][0] - res["fp32"][0]).abs().max() / res["fp32"][0].abs().max())
    e_p = float((res[mode][1] - res["fp32"][1]).abs().max() / res["fp32"][1].abs().max())
    same_argmax = bool((res[mode][1].flatten(1, 3).argmax(1) == res["fp32"][1].flatten(1, 3).argmax(1)).all())
    print("%-7s vs fp32 mode: TCNet.forward %.2e, attention map %.2e, argmax identical: %s" % (mode, e_raw, e_p, same_argmax))
