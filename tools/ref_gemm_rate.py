#!/usr/bin/env python3
"""Reference point for the plain-bf16 projection GEMM (NOT on the product path): what the vendor library behind torch.matmul reaches on the same MI355X at the
shapes of the hoisted projections (bf16 in, fp32 accumulate), next to this library's `cti_gemm_nt_pb` in the plain-bf16 mode (tools/bench_gemm_pb.py).
python tools/ref_gemm_rate.py [reps]"""
import json, sys
import torch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
res = {}
for M, N, K in [(9216, 3072, 2048), (9216, 11264, 2048), (9216, 1024, 2048), (3584, 3072, 1024), (256, 1024, 1024)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): torch.matmul(a, b.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): torch.matmul(a, b.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    res["%dx%dx%d" % (M, N, K)] = [round(us, 1), round(2.0 * M * N * K / us * 1e-6, 1)]
print(json.dumps({"what": "torch.matmul bf16 (vendor GEMM), us and TFLOP/s", "us_tflops": res}))
