#!/usr/bin/env python3
"""A/B timing of the M-build forward / backward kernels across library variants built by `tools/tune_gemm.py build name:-Dflags ...`
(interleaved rounds in one process).  Shapes: B=256, V=36, Q=14, R=32, hr=16, G=2 (every model configuration's TriAttention).
    python tools/tune_mbuild.py [rounds]"""
import ctypes as C
import glob
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "iccv19_vqa-cti_amd", "lib", "variants")


def main(rounds=5):
    import torch
    import cti_amd
    L = cti_amd.pkg._lib
    libs = {}
    for f in sorted(glob.glob(os.path.join(VDIR, "*.so"))):
        l = C.CDLL(f)
        for name, (res, args) in L.SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        libs[os.path.basename(f)[len("libcti_hip_"):-3]] = l
    B, V, Q, R, hr, G = 256, 36, 14, 32, 16, 2
    K = R * hr
    dev = "cuda"
    torch.manual_seed(0)
    Vr = torch.randn(B, V, K, device=dev); Qr = torch.randn(B, Q, K, device=dev)
    Te = torch.randn(R, hr, hr, hr, G, device=dev)
    dM = torch.randn(B, V, Q, G, K, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    ref = None
    for name, l in libs.items():
        dVr = torch.zeros_like(Vr); dQr = torch.zeros_like(Qr); dT = torch.zeros(B, R, hr, hr, hr, G, device=dev)
        M = torch.zeros(B, V, Q, G, K, device=dev)

        def bwd(l=l, dVr=dVr, dQr=dQr, dT=dT):
            rc = l.cti_paralind_mbuild_bwd(dM.data_ptr(), Vr.data_ptr(), Qr.data_ptr(), Te.data_ptr(), dVr.data_ptr(), dQr.data_ptr(), dT.data_ptr(),
                                           B, V, Q, R, hr, G, st)
            assert rc == 0, l.cti_last_error_string()

        def fwd(l=l, M=M):
            rc = l.cti_paralind_mbuild_fwd(Vr.data_ptr(), Qr.data_ptr(), Te.data_ptr(), M.data_ptr(), B, V, Q, R, hr, hr, hr, G, st)
            assert rc == 0, l.cti_last_error_string()
        rows_alloc = B * V * Q * G + 256
        Mh = torch.zeros(K // 16, rows_alloc, 16, device=dev, dtype=torch.int16); Ml = torch.zeros_like(Mh)
        Tt = Te.view(R, hr, hr * hr * G).transpose(1, 2).contiguous()

        def planes(l=l, Mh=Mh, Ml=Ml, Tt=Tt):
            rc = l.cti_paralind_mbuild_planes_fwd(Vr.data_ptr(), Qr.data_ptr(), Te.data_ptr(), Tt.data_ptr(), Mh.data_ptr(), Ml.data_ptr(), B, V, Q, R, hr, G,
                                                  rows_alloc, st)
            assert rc == 0, l.cti_last_error_string()
        bwd(); fwd(); planes(); torch.cuda.synchronize()
        # planes -> fp32 (hi + lo) in the (B,V,Q,G,K) layout, against the fp32 M of the same library
        def unplane(P):
            x = (P.to(torch.int32) & 0xFFFF) << 16
            return x.view(torch.float32)[:, :B * V * Q * G, :].permute(1, 0, 2).reshape(B, V, Q, G, K)
        Mp = unplane(Mh) + unplane(Ml)
        print("%-12s planes vs fp32 M: %.2e" % (name, float((Mp - M).abs().max() / M.abs().max())))
        outs = [dVr.clone(), dQr.clone(), dT.sum(0), M.clone()]
        if ref is None:
            ref = outs
        else:
            print("%-12s max diff vs first: dVr %.2e dQr %.2e dT %.2e M %.2e" % ((name,) + tuple(float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs, ref))))
        res[name] = (bwd, fwd, planes, [], [], [])
    for _ in range(rounds):
        for name, (bwd, fwd, planes, tb, tf, tp) in res.items():
            for fn, acc in ((bwd, tb), (fwd, tf), (planes, tp)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record(); torch.cuda.synchronize()
                acc.append(e0.elapsed_time(e1) / 5)
    for name, (_, _, _, tb, tf, tp) in res.items():
        print("%-12s bwd median %.3f ms (min %.3f)   fwd median %.3f ms (min %.3f)   planes fwd median %.3f ms (min %.3f)"
              % (name, statistics.median(tb), min(tb), statistics.median(tf), min(tf), statistics.median(tp), min(tp)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
