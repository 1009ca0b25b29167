#!/usr/bin/env python3
"""The plain-bf16 projection GEMM (csrc/cti_gemm16.hip) at the hoisted projections' shapes, next to the vendor GEMM behind torch.matmul (yardstick, not
on the product path), interleaved in ONE process on the same random operands:
  rows_f32 / rows_bf16   cti_gemm_bf16_rows: A a row-major bf16 matrix read as it stands, fp32 / bf16 rows out (bias + ReLU in the epilogue)
  rows_bf16_whole_tiles  the same without the round-6 stream-K cut (cti_gemm_bf16_rows)
  f32_in                 cti_gemm_nt_pb in the bf16 mode: fp32 A -> hi plane (one pass) -> product (conversion INCLUDED)
  vendor                 torch.matmul(a_bf16, w_bf16.t()) (bf16 out, no epilogue)
python tools/bench_gemm16.py [reps]"""
import json, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops = cti_amd.pkg.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(0)
res = {}


def timeit(fns, reps):
    out = {k: [] for k in fns}
    for k, f in fns.items():
        for _ in range(3): f()
    torch.cuda.synchronize()
    for rnd in range(3):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            out[k].append(e0.elapsed_time(e1) / reps * 1e3)
    return {k: min(v) for k, v in out.items()}


for M, N, K in [(9216, 3072, 2048), (9216, 8192, 2048), (9216, 11264, 2048), (9216, 1024, 2048), (3584, 3072, 1024), (4096, 4096, 4096)]:
    a32 = torch.randn(M, K, generator=g).cuda(); w32 = (torch.randn(N, K, generator=g) / 8).cuda(); b = torch.randn(N, generator=g).cuda()
    a16, w16 = a32.to(torch.bfloat16), w32.to(torch.bfloat16)
    wp = ops.split_operand(w32, prec="bf16")
    fns = {
        "rows_f32": lambda: ops.gemm_bf16_rows(a16, wp, N, bias=b, relu=True),
        "rows_bf16": lambda: ops.gemm_bf16_rows(a16, wp, N, out_dtype=torch.bfloat16, bias=b, relu=True),
        "rows_bf16_whole_tiles": lambda: ops.gemm_bf16_rows(a16, wp, N, out_dtype=torch.bfloat16, bias=b, relu=True, stream_k=False),
        "f32_in": lambda: ops.gemm_nt(a32, w32, bias=b, relu=True, prec="bf16", B_planes=wp),
        "vendor": lambda: torch.matmul(a16, w16.t()),
    }
    t = timeit(fns, reps)
    res["%dx%dx%d" % (M, N, K)] = {k: [round(us, 1), round(2.0 * M * N * K / us * 1e-6, 1)] for k, us in t.items()}
print(json.dumps({"what": "us and TFLOP/s (min of 3 interleaved rounds)", "shapes": res}))
