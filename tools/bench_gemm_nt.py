#!/usr/bin/env python3
"""Times ops.gemm_nt (x @ w^T through the plane GEMM) at the models' shapes.  With CTI_HIP_LIB pointing at builds made with
-DCTI_FORCE_CFG=0/1/2 (128x128 / 256x128 / 256x256 tiles) this is the measurement behind the tile-choice model of gemm_nt_planes.
python tools/bench_gemm_nt.py"""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops = cti_amd.pkg.ops
shapes = [(9216, 512, 2048), (9216, 1024, 2048), (9216, 3072, 2048), (9216, 2048, 1024), (9216, 512, 512), (3584, 512, 1024), (3584, 1024, 1024),
          (3584, 3072, 1024), (768, 512, 1024), (768, 1024, 1024), (256, 1024, 1024), (256, 3072, 1024), (256, 2048, 1024), (256, 3129, 2048),
          (2304, 1024, 2048), (4608, 1024, 1024), (18432, 256, 512)]
g = torch.Generator().manual_seed(0)
res = {}
for M, N, K in shapes:
    a = torch.randn(M, K, generator=g).cuda(); b = torch.randn(N, K, generator=g).cuda()
    for _ in range(3): ops.gemm_nt(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_nt(a, b)
    e1.record(); torch.cuda.synchronize()
    res["%dx%dx%d" % (M, N, K)] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
print(json.dumps({"lib": os.path.basename(os.environ.get("CTI_HIP_LIB", "default")), "us": res}))
