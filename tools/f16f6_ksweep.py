#!/usr/bin/env python3
"""Time cti_gemm_nt_f16f6 at the BASELINE configs[1] mode-3 shape for several K: the slope is the cost of one 32-deep K block per tile, the
intercept the per-tile cost (epilogue stores, tile set-up, ring fill).   python tools/f16f6_ksweep.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd  # noqa: E402

ops, L = cti_amd.ops, cti_amd.pkg._lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
V, Q, A, G = 36, 14, 3129, 2
dev = "cuda"
lib = L.lib()
out = torch.empty((B, V * Q, A, G), device=dev)
st = ops._stream()
tiles = B * 4 * 17
for K in (32, 64, 128, 256, 512, 1024):
    g = torch.Generator(device=dev).manual_seed(1)
    M = torch.randn(B * V * Q * G, K, device=dev, generator=g)
    Ar = torch.randn(B * A, K, device=dev, generator=g)
    pa, pb = ops.quantize_f16f6(M, V * Q * G), ops.quantize_f16f6(Ar, A)
    del M, Ar

    def f6():
        L.check(lib.cti_gemm_nt_f16f6(pa.data_ptr(), B * V * Q * G, V * Q * G, pb.data_ptr(), B * A, A, out.data_ptr(), A * G, G, V * Q * A * G, G, B,
                                      V * Q * G, A, K, 0, 1, 0, 0, st), "f16f6")
    for _ in range(3):
        f6()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f6()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("K=%4d  %.3f ms   per tile and CU %.2f us   (%d K blocks)" % (K, ms, ms * 1e3 / (tiles / 256), K // 32))
    del pa, pb

# the output stream alone, for scale: torch's fill of the same 3.2 GB buffer
for _ in range(2):
    out.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out.zero_()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("fill of the %.2f GB output: %.3f ms = %.2f TB/s" % (out.numel() * 4 / 1e9, ms, out.numel() * 4 / ms / 1e9))
