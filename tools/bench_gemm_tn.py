#!/usr/bin/env python3
"""Times ops.gemm_tn (the weight-gradient contraction a^T b over the rows) at the training shapes; CTI_TN_PLAN=0 selects the first split
planner for an A/B in a second process.  python tools/bench_gemm_tn.py"""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops = cti_amd.pkg.ops
shapes = [(9216, 3072, 2048), (9216, 1024, 2048), (9216, 1024, 1024), (9216, 512, 2048), (9216, 512, 512), (3584, 1024, 1024), (3584, 512, 1024),
          (3584, 3072, 1024), (768, 1024, 1024), (256, 2048, 1024), (256, 3129, 2048)]
g = torch.Generator().manual_seed(0)
res = {}
for M, N, K in shapes:
    a = torch.randn(M, N, generator=g).cuda(); b = torch.randn(M, K, generator=g).cuda()
    a = torch.relu(a)                                      # like dzs: about half zeros
    for _ in range(3): ops.gemm_tn(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_tn(a, b)
    e1.record(); torch.cuda.synchronize()
    res["%dx%dx%d" % (M, N, K)] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
print(json.dumps({"plan": os.environ.get("CTI_TN_PLAN", "1"), "us": res}))
