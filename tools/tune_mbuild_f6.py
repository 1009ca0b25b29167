#!/usr/bin/env python3
"""A/B of cti_paralind_mbuild_f16f6_fwd (the M build that encodes the mode-3 product's f16f6 planes itself) at the BASELINE configs[1] shape:
one small library per -D flag set (cti_mbuild.hip + cti_paralind.hip + cti_api.hip, -Bsymbolic; the rest resolves against the main library),
interleaved rounds in one process.  Also times the pair it replaces (fp32 M build is not exported: the encoder pass alone is shown).
    python tools/tune_mbuild_f6.py build base: early0:-DCTI_MBF6_PREFETCH_EARLY=0 ...   (here)
    python tools/tune_mbuild_f6.py run [rounds]                                          (GPU box)
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
        out = os.path.join(VDIR, "libmbf6_%s.so" % name)
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-Wno-unused-result"] + flags.split() + ["-o", out] + [
            os.path.join(CSRC, f) for f in ("cti_mbuild.hip", "cti_paralind.hip", "cti_api.hip")]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for n, p in procs:
        o, _ = p.communicate()
        print(n, "rc", p.returncode, o[-600:] if p.returncode else "")


def run(rounds=5):
    import torch
    import cti_amd
    L, ops = cti_amd.pkg._lib, cti_amd.ops
    C.CDLL(os.path.join(ROOT, "iccv19_vqa-cti_amd", "lib", "libcti_hip.so"), mode=C.RTLD_GLOBAL)
    libs = {}
    for f in sorted(glob.glob(os.path.join(VDIR, "libmbf6_*.so")), key=lambda x: (not os.path.basename(x).startswith("libmbf6_base"), x)):
        l = C.CDLL(f)
        for name in ("cti_paralind_mbuild_f16f6_fwd", "cti_last_error_string"):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = L.SIGNATURES[name]
        libs[os.path.basename(f)[len("libmbf6_"):-3]] = l
    dev = "cuda"
    B, V, Q, R, hr, G = 256, 36, 14, 32, 16, 2
    Vr = torch.relu(torch.randn(B, V, R * hr, device=dev))
    Qr = torch.relu(torch.randn(B, Q, R * hr, device=dev))
    Teff = torch.randn(R, hr, hr, hr, G, device=dev)
    Tt = ops.transpose(Teff, hr, hr * hr * G, batch=R, s_src=hr ** 3 * G, ld_src=hr * hr * G, s_dst=hr ** 3 * G, ld_dst=hr).view(R, hr * hr * G, hr)
    nb = L.lib().cti_f16f6_planes_bytes(B * V * Q * G, R * hr, V * Q * G)
    blk = torch.zeros(nb, device=dev, dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    times = {k: [] for k in libs}
    ref = None
    for rnd in range(rounds + 1):
        for name, l in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                rc = l.cti_paralind_mbuild_f16f6_fwd(Vr.data_ptr(), Qr.data_ptr(), Tt.data_ptr(), blk.data_ptr(), nb, B, V, Q, R, hr, G, st)
                assert rc == 0, (name, l.cti_last_error_string())
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[name].append(e0.elapsed_time(e1) / 5)
            elif ref is None:
                ref = blk.clone()
            else:
                print("%-12s planes identical to the first variant: %s" % (name, bool(torch.equal(blk, ref))))
    for name, ts in times.items():
        print("%-14s median %.3f ms (min %.3f)" % (name, statistics.median(ts), min(ts)))
    M = torch.randn(B * V * Q * G, R * hr, device=dev)
    for _ in range(2):
        ops.quantize_f16f6(M, V * Q * G)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.quantize_f16f6(M, V * Q * G)
    e1.record(); torch.cuda.synchronize()
    print("encoder pass over an fp32 M (what the direct build removes, beside the fp32 write): %.3f ms" % (e0.elapsed_time(e1) / 5))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
