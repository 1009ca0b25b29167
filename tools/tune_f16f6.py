#!/usr/bin/env python3
"""A/B tuner for the f16f6 GEMM (cti_gemm_f16f6.hip): builds one libcti_hip_<name>.so per -D flag set here, then on the GPU loads them all in
ONE process and times cti_gemm_nt_f16f6 at the BASELINE configs[1] mode-3 shape in interleaved rounds.
    python tools/tune_f16f6.py build name1:-DFOO=1 name2:"-DFOO=2 -DBAR" ...      (here; the .so files travel with gpurun)
    python tools/tune_f16f6.py run [rounds] [B] [K,K,..]                                (on the GPU box)
"""
import ctypes as C
import glob
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
VDIR = os.path.join(ROOT, "iccv19_vqa-cti_amd", "lib", "variants")


def run(rounds=5, B=256, K=512):
    import torch
    import cti_amd
    L, ops = cti_amd.pkg._lib, cti_amd.ops
    libs = {}
    C.CDLL(os.path.join(ROOT, "iccv19_vqa-cti_amd", "lib", "libcti_hip.so"), mode=C.RTLD_GLOBAL)
    for f in sorted(glob.glob(os.path.join(VDIR, "libcti_hip_*.so")), key=lambda x: (not os.path.basename(x).startswith("libcti_hip_base"), x)):
        l = C.CDLL(f)
        for name in ("cti_gemm_nt_f16f6", "cti_last_error_string"):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = L.SIGNATURES[name]
        libs[os.path.basename(f)[len("libcti_hip_"):-3]] = l
    V, Q, A, G = 36, 14, 3129, 2
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    M = torch.randn(B * V * Q * G, K, device=dev, generator=g) * 8.0
    Ar = torch.relu(torch.randn(B * A, K, device=dev, generator=g) * 0.7)
    pa, pb = ops.quantize_f16f6(M, V * Q * G), ops.quantize_f16f6(Ar, A)
    out = torch.empty((B, V * Q, A + 256, G), device=dev)             # (room for the padded-row store ablation)
    st = torch.cuda.current_stream().cuda_stream
    times = {k: [] for k in libs}
    ref = None
    for rnd in range(rounds + 1):
        for name, l in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rc = l.cti_gemm_nt_f16f6(pa.data_ptr(), M.shape[0], V * Q * G, pb.data_ptr(), Ar.shape[0], A, out.data_ptr(), A * G, G, V * Q * A * G, G, B,
                                         V * Q * G, A, K, 0, 1, 0, 0, st)
                assert rc == 0, (name, l.cti_last_error_string())
            e1.record(); torch.cuda.synchronize()
            if name.startswith("clk") and rnd == rounds:               # -DCTI_F6_ABL=128 variants: per-workgroup (shader cycles, 100 MHz ticks)
                pr = out.view(-1)[:512].view(256, 2).double()
                print("K=%d %-12s shader clock %.0f MHz (min %.0f max %.0f), workgroup time %.3f ms" % (
                    K, name, (pr[:, 0] / pr[:, 1] * 100).mean().item(), (pr[:, 0] / pr[:, 1] * 100).min().item(), (pr[:, 0] / pr[:, 1] * 100).max().item(), pr[:, 1].mean().item() / 1e5))
            if rnd:
                times[name].append(e0.elapsed_time(e1) / 3)
            elif ref is None:
                ref = out.view(-1)[:2 * V * Q * A * G].clone()
            else:
                print("%-12s max diff vs first variant: %.2e" % (name, (out.view(-1)[:2 * V * Q * A * G] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)))
    flops = 2.0 * B * V * Q * G * A * K
    for name, ts in times.items():
        print("K=%d %-12s median %.3f ms (min %.3f)  %.0f TFLOP/s" % (K, name, statistics.median(ts), min(ts), flops / statistics.median(ts) / 1e9))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        # light variants: the GEMM's own source + cti_api.hip, -Bsymbolic (internal calls stay inside the variant), everything else resolves
        # against the main library, which run() loads RTLD_GLOBAL first
        import subprocess
        os.makedirs(VDIR, exist_ok=True)
        for f in glob.glob(os.path.join(VDIR, "*.so")):
            os.remove(f)
        procs = []
        for spec in sys.argv[2:]:
            name, _, flags = spec.partition(":")
            cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-Wno-unused-result"] + flags.split() + [
                "-o", os.path.join(VDIR, "libcti_hip_%s.so" % name)] + [os.path.join(ROOT, "iccv19_vqa-cti_amd", "csrc", f) for f in ("cti_gemm_f16f6.hip", "cti_api.hip")]
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for n, pr in procs:
            o, _ = pr.communicate()
            print(n, "rc", pr.returncode, o[-600:] if pr.returncode else "")
    else:
        for K in ([int(k) for k in sys.argv[4].split(",")] if len(sys.argv) > 4 else [512]):
            run(int(sys.argv[2]) if len(sys.argv) > 2 else 5, int(sys.argv[3]) if len(sys.argv) > 3 else 256, K)
