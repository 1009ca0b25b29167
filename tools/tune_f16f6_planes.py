#!/usr/bin/env python3
"""A/B of cti_gemm_nt_f16f6_planes (the a-side rank nets of the f16f6 mode: planes -> planes, transposed product, register epilogue) at the
BASELINE configs[1] shape: one small library per -D flag set (only cti_gemm_f16f6.hip + cti_api.hip), interleaved rounds in one process.
    python tools/tune_f16f6_planes.py build base: nostore:-DCTI_F6_ABL=8 ...     (here)
    python tools/tune_f16f6_planes.py run [rounds]                                (GPU box)
"""
import ctypes as C
import glob
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "iccv19_vqa-cti_amd", "lib", "variants")
CSRC = os.path.join(ROOT, "iccv19_vqa-cti_amd", "csrc")


def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    for f in glob.glob(os.path.join(VDIR, "*.so")):
        os.remove(f)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(VDIR, "libf6p_%s.so" % name)
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-Wno-unused-result"] + flags.split() + ["-o", out,
               os.path.join(CSRC, "cti_gemm_f16f6.hip"), os.path.join(CSRC, "cti_api.hip")]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for n, p in procs:
        o, _ = p.communicate()
        print(n, "rc", p.returncode, o[-600:] if p.returncode else "")


def run(rounds=5):
    import torch
    import cti_amd
    L, ops = cti_amd.pkg._lib, cti_amd.ops
    libs = {}
    C.CDLL(L.LIB_PATH if hasattr(L, "LIB_PATH") else os.path.join(ROOT, "iccv19_vqa-cti_amd", "lib", "libcti_hip.so"), mode=C.RTLD_GLOBAL)   # the variants hold two sources only
    for f in sorted(glob.glob(os.path.join(VDIR, "libf6p_*.so")), key=lambda x: (not os.path.basename(x).startswith("libf6p_base"), x)):
        l = C.CDLL(f)
        for name in ("cti_gemm_nt_f16f6_planes", "cti_last_error_string"):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = L.SIGNATURES[name]
        libs[os.path.basename(f)[len("libf6p_"):-3]] = l
    dev = "cuda"
    rows, K, M, A = 256 * 3129, int(os.environ.get("CTI_TUNE_K", "512")), 512, 3129      # CTI_TUNE_K=300: the Tucker projection's shape
    x = torch.relu(torch.randn(rows, K, device=dev))
    w = torch.randn(M, K, device=dev) * 0.05
    sc = torch.rand(32, device=dev) + 0.5
    bi = torch.randn(M, device=dev)
    px, pw = ops.quantize_f16f6(x), ops.quantize_f16f6(w)
    nby = L.lib().cti_f16f6_planes_bytes(rows, M, A)
    y = torch.zeros(nby, device=dev, dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    times = {k: [] for k in libs}
    for rnd in range(rounds + 1):
        for name, l in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rc = l.cti_gemm_nt_f16f6_planes(pw.data_ptr(), M, px.data_ptr(), rows, y.data_ptr(), nby, A, M, rows, K, bi.data_ptr(), 1, st)
                assert rc == 0, (name, l.cti_last_error_string())
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[name].append(e0.elapsed_time(e1) / 3)
    flops = 2.0 * rows * M * K
    for name, ts in times.items():
        print("%-14s median %.3f ms (min %.3f)  %.0f TFLOP/s" % (name, statistics.median(ts), min(ts), flops / statistics.median(ts) / 1e9))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
