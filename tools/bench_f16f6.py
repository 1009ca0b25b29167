#!/usr/bin/env python3
"""Mode-3 GEMM of BASELINE configs[1] (256 x [1008 x 512] . [3129 x 512]^T, rows interleaved by G = 2) alone: the f16 + fp6 split product
(cti_gemm_nt_f16f6 on pre-encoded planes) against the bf16x3 plane GEMM, same random operands, interleaved launches; error of each against
float64 on one sample.   python tools/bench_f16f6.py [B] [reps]"""
import os
import sys
import json

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd  # noqa: E402

ops, L = cti_amd.ops, cti_amd.pkg._lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
V, Q, A, G, K = 36, 14, 3129, 2, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
M = torch.randn(B * V * Q * G, K, device=dev, generator=g) * 8.0
Ar = torch.relu(torch.randn(B * A, K, device=dev, generator=g) * 0.7)
lib = L.lib()
pa = ops.quantize_f16f6(M, V * Q * G)
pb = ops.quantize_f16f6(Ar, A)
out = torch.empty((B, V * Q, A, G), device=dev)
st = ops._stream()


def f6():
    L.check(lib.cti_gemm_nt_f16f6(pa.data_ptr(), M.shape[0], V * Q * G, pb.data_ptr(), Ar.shape[0], A, out.data_ptr(), A * G, G, V * Q * A * G, G, B,
                                  V * Q * G, A, K, 0, 1, 0, 0, st), "f16f6")


Mh, Ml = None, None
M5 = M.view(B, V, Q, G, K)


def x3():
    return ops.paralind_core(M5, Ar.view(B, A, K), prec="bf16x3")


def timeit(fn, n):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


f6()
o6 = out[0].clone()
o3 = x3()[0].reshape(V * Q, A, G).clone()
ref = (M[:V * Q * G].double() @ Ar[:A].double().T).view(V * Q, G, A).permute(0, 2, 1)
nrm = ref.abs().max()
e6 = float((o6.double() - ref).abs().max() / nrm)
e3 = float((o3.double() - ref).abs().max() / nrm)
t6 = timeit(f6, reps)
ops.profile_start()
for _ in range(5):
    x3()
kt = ops.profile_stop()
flops = 2.0 * B * V * Q * G * A * K
print(json.dumps({"B": B, "f16f6_ms": round(t6, 4), "f16f6_tflops": round(flops / t6 / 1e9, 1), "f16f6_err": e6, "bf16x3_err": e3,
                  "bf16x3_core_ms_incl_split": round(float(np.mean(kt["paralind_core"][1:])), 4)}))
