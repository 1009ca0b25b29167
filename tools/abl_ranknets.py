#!/usr/bin/env python3
"""Times cti_ranknets_drop_fwd / _dw / _dx alone at the visual-branch shape (rows 9216, h 512, R 32, hr 16); CTI_HIP_LIB selects another build of
the library (kernel variants are A/B-ed this way; the forward kernel's comment in csrc/cti_ranknets.hip lists what was tried).
python tools/abl_ranknets.py [rows]"""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops = cti_amd.pkg.ops
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 9216
h, R, hr, p = 512, 32, 16, 0.5
g = torch.Generator().manual_seed(1)
x = torch.randn(rows, h, generator=g).cuda()
W = (torch.randn(R * hr, h, generator=g) / 16).cuda()
scale = torch.rand(R, generator=g).cuda() + 0.5
bias = torch.randn(R * hr, generator=g).cuda()
dzs = torch.randn(rows, R * hr, generator=g).cuda()
mask = ops.dropout_mask((R, rows, h), p, x.device)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = {"lib": os.environ.get("CTI_HIP_LIB", "default"), "rows": rows,
       "fwd_us": round(t(lambda: ops.ranknets_drop_fwd(x, mask, W, scale, bias, R, p, True)), 1),
       "dw_us": round(t(lambda: ops.ranknets_drop_dw(dzs, x, mask, R, p)), 1),
       "dx_us": round(t(lambda: ops.ranknets_drop_dx(dzs, W, mask, R, p)), 1)}
print(json.dumps(out))
