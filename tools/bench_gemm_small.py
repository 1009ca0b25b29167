#!/usr/bin/env python3
"""The mid-size plain-bf16 products of the BASELINE configs[2] / [3] model forwards (shapes from CTI_GEMM_TRACE=1): split of x + product against resident weight planes
(`cti_gemm_nt_pb`), scale + bias epilogue.  CTI_GEMM16_SMALL=0 keeps the 128 x 128 / 256 x 128 tiles on the planes kernel of cti_gemm_bf16x3.hip (A/B).
python tools/bench_gemm_small.py [reps]"""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops = cti_amd.pkg.ops
cti_amd.set_precision(os.environ.get("CTI_PREC", "bf16"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(0)
res = {}
for M, n, N, K in [(3072, 2, 1024, 1024), (1536, 2, 1024, 1024), (3584, 8, 1024, 1024), (3584, 2, 1024, 1024), (2304, 1, 512, 512), (1536, 1, 512, 512), (9216, 1, 512, 512), (3584, 1, 512, 512),
                   (768, 1, 512, 512), (3072, 1, 3072, 608), (1536, 1, 3072, 608), (3584, 1, 3072, 1024)]:
    a = torch.randn(M, K, generator=g).cuda(); w = torch.randn(n * N, K, generator=g).cuda() / K ** 0.5
    bias = torch.randn(n * N, generator=g).cuda()
    wp = ops.split_operand(w)
    f = lambda: ops.gemm_nt(a, w, nb1=n, rA1=0, rB1=N, M=M, N=N, bias=bias, bias_bs=N, B_planes=wp)    # noqa: E731
    out = f()
    ref = (a.double() @ w.double().t() + bias.double()).view(M, n, N).permute(1, 0, 2)
    err = float((out.double().reshape(ref.shape) - ref).abs().max() / ref.abs().max())
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    us = sorted(ts)[1]
    res["%dx(%dx%d)x%d" % (M, n, N, K)] = [round(us, 1), round(2.0 * M * n * N * K / us * 1e-6, 1), "%.1e" % err]
print(json.dumps({"small16": os.environ.get("CTI_GEMM16_SMALL", "1"), "prec": cti_amd.get_precision(), "us_tflops_err (split of x included)": res}))
