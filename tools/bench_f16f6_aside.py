#!/usr/bin/env python3
"""a-side shapes of BASELINE configs[1] on the f16f6 GEMM (fp32-out epilogue, pre-encoded planes) and the encoder alone."""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops, L = cti_amd.ops, cti_amd.pkg._lib
lib = L.lib()
dev = "cuda"
rows = 256 * 3129
st = ops._stream()


def t(fn, n=5):
    for _ in range(2):
        fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); e[0].record()
    for _ in range(n):
        fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n


for K in (300, 512):
    x = torch.randn(rows, K, device=dev)
    w = torch.randn(512, K, device=dev)
    tq = t(lambda: ops.quantize_f16f6(x))
    px, pw = ops.quantize_f16f6(x), ops.quantize_f16f6(w)
    out = torch.empty(rows, 512, device=dev)
    tg = t(lambda: L.check(lib.cti_gemm_nt_f16f6(px.data_ptr(), rows, 0, pw.data_ptr(), 512, 0, out.data_ptr(), 512, 1, 0, 1, 1, rows, 512, K, 0, 1, 0, 0, st), "g"))
    print(json.dumps({"K": K, "quantize_ms": round(tq, 3), "gemm_f32out_ms": round(tg, 3), "gemm_tflops": round(2.0 * rows * 512 * K / tg / 1e9, 1)}))

# the a-side rank nets as they run in the f16f6 mode: planes -> planes, transposed product, register epilogue
K, M = 512, 512
x = torch.relu(torch.randn(rows, K, device=dev))
w = torch.randn(M, K, device=dev) * 0.05
sc = torch.rand(32, device=dev) + 0.5
bi = torch.randn(M, device=dev)
px, pw = ops.quantize_f16f6(x), ops.quantize_f16f6(w)
nby = lib.cti_f16f6_planes_bytes(rows, M, 3129)
y = torch.zeros(nby, device=dev, dtype=torch.uint8)
tp = t(lambda: L.check(lib.cti_gemm_nt_f16f6_planes(pw.data_ptr(), M, px.data_ptr(), rows, y.data_ptr(), nby, 3129, M, rows, K, bi.data_ptr(), 1, st), "p"), n=10)
print(json.dumps({"planes_to_planes_ms": round(tp, 3), "tflops": round(2.0 * rows * M * K / tp / 1e9, 1)}))
