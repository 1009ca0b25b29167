#!/usr/bin/env python3
"""The hoisted projection GEMMs of the model forwards (x (9216, 2048) against n batched (1024, 2048) weights with resident planes: split of x + product,
`cti_gemm_nt_pb`) in the precision of CTI_PREC (default bf16).  CTI_HIP_LIB selects a variant build (tools/tune_gemm.py build ...).
python tools/bench_gemm_pb.py [reps]"""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops = cti_amd.pkg.ops
cti_amd.set_precision(os.environ.get("CTI_PREC", "bf16"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(0)
res = {}
for M, n, N, K in [(9216, 1, 3072, 2048), (9216, 3, 1024, 2048), (9216, 11, 1024, 2048), (9216, 1, 1024, 2048), (3584, 1, 3072, 1024), (2304, 3, 1024, 2048)]:
    a = torch.randn(M, K, generator=g).cuda(); w = torch.randn(n * N, K, generator=g).cuda()
    wp = ops.split_operand(w)
    f = lambda: ops.gemm_nt(a, w, nb1=n, rA1=0, rB1=N, M=M, N=N, relu=True, B_planes=wp)    # noqa: E731
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    res["%dx(%dx%d)x%d" % (M, n, N, K)] = [round(us, 1), round(2.0 * M * n * N * K / us * 1e-6, 1)]
print(json.dumps({"lib": os.path.basename(os.environ.get("CTI_HIP_LIB", "default")), "prec": cti_amd.get_precision(), "us_tflops": res}))
