#!/usr/bin/env python3
"""A/B tuner for kernel variants: builds one libcti_hip_<name>.so per -D flag set (locally, with hipcc), then on the GPU
loads them all in ONE process and times cti_tcnet_forward at the BASELINE configs[1] shapes in interleaved rounds
(cdna_hip_programming.md rule 24).  Usage:
    python tools/tune_gemm.py build  name1:-DFOO=1  name2:-DFOO=2 ...     (here; .so files travel with gpurun)
    python tools/tune_gemm.py run [rounds]                                 (on the GPU box)
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


def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    for f in glob.glob(os.path.join(VDIR, "*.so")):
        os.remove(f)
    srcs = sorted(glob.glob(os.path.join(ROOT, "iccv19_vqa-cti_amd", "csrc", "*.hip")))
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(VDIR, "libcti_hip_%s.so" % name)
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-unused-result"] + flags.split() + ["-o", out] + srcs
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) >= 4:
            for n, p in procs:
                o, _ = p.communicate()
                print(n, "rc", p.returncode, o[-400:] if p.returncode else "")
            procs = []
    for n, p in procs:
        o, _ = p.communicate()
        print(n, "rc", p.returncode, o[-400:] if p.returncode else "")


def run(rounds=5, prec=1):
    import torch
    import bench
    import cti_amd
    L = cti_amd.pkg._lib
    libs = {}
    for f in sorted(glob.glob(os.path.join(VDIR, "*.so"))):
        l = C.CDLL(f)
        for name, (res, args) in L.SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:                   # a variant built from an older commit: symbols added since are not needed here
                continue
            fn.restype, fn.argtypes = res, args
        libs[os.path.basename(f)[len("libcti_hip_"):-3]] = l
    c = dict(bench.C2)
    dev = torch.device("cuda:0")
    torch.manual_seed(1204)
    net = cti_amd.TCNet(c["v_dim"], c["q_dim"], c["a_dim"], c["h_mm"], 1, c["rank"], c["glimpse"]).to(dev).eval()
    v, q, a = bench.synth_inputs(c, c["B"], 1205, dev)
    tucker, rank = net._fused_args()
    B, V, Q, A, G, h, R = c["B"], c["V"], c["Q"], c["A"], c["glimpse"], c["h_mm"], c["rank"]
    arr = lambda ts: (C.c_void_p * 3)(*[t.contiguous().data_ptr() for t in ts])
    keep = [t.contiguous() for tr in (tucker, rank) for tt in tr for t in tt]
    args6 = [arr([t[i] for t in tucker]) for i in range(3)] + [arr([t[i] for t in rank]) for i in range(3)]
    Tg = net.T_g.detach().contiguous()
    out = torch.empty((B, V, Q, A, G), device=dev)
    ref = None
    times = {k: ([], []) for k in libs}
    st = torch.cuda.current_stream().cuda_stream
    # the auxiliary stream of cti_tcnet_forward: on by default like the product path (CTI_TUNE_AUX=0 disables, CTI_TUNE_AUX_PRIO=-1 = high priority)
    aux_obj = torch.cuda.Stream(priority=int(os.environ.get('CTI_TUNE_AUX_PRIO', '0'))) if os.environ.get('CTI_TUNE_AUX', '1') == '1' else None
    AUX = aux_obj.cuda_stream if aux_obj is not None else None
    use_prep = os.environ.get('CTI_TUNE_PREPARED', '1') == '1'          # weights held as a prepared block, like TCNet in eval mode
    preps = {}
    for name, l in libs.items():
        if use_prep and hasattr(l, "cti_tcnet_prepare"):
            nb = l.cti_tcnet_prepared_bytes(c["v_dim"], c["q_dim"], c["a_dim"], h, R, G, prec)
            blk = torch.empty(nb, device=dev, dtype=torch.uint8)
            rc = l.cti_tcnet_prepare(args6[0], args6[1], args6[3], args6[4], Tg.data_ptr(), c["v_dim"], c["q_dim"], c["a_dim"], h, R, G, prec, blk.data_ptr(), nb, st)
            assert rc == 0, l.cti_last_error_string()
            preps[name] = blk
    for rnd in range(rounds + 1):
        for name, l in libs.items():
            wsb = l.cti_tcnet_forward_workspace_bytes(B, V, Q, A, c["v_dim"], c["q_dim"], c["a_dim"], h, R, G, prec)
            ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
            e = [l.cti_event_create() for _ in range(4)]
            l.cti_event_record(e[0], st)
            rc = l.cti_tcnet_forward(v.data_ptr(), q.data_ptr(), a.data_ptr(), *args6, Tg.data_ptr(), out.data_ptr(), None, B, V, Q, A,
                                     c["v_dim"], c["q_dim"], c["a_dim"], h, R, G, 1, prec, preps[name].data_ptr() if name in preps else None,
                                     ws.data_ptr(), wsb, e[1], e[2], AUX, st)
            assert rc == 0, (name, rc, l.cti_last_error_string())
            l.cti_event_record(e[3], st)
            torch.cuda.synchronize()
            ms = C.c_float()
            l.cti_event_elapsed_ms(e[1], e[2], C.byref(ms)); core = ms.value
            l.cti_event_elapsed_ms(e[0], e[3], C.byref(ms)); tot = ms.value
            for x in e:
                l.cti_event_destroy(x)
            if rnd:
                times[name][0].append(core); times[name][1].append(tot)
            else:
                if ref is None:
                    ref = out.clone()
                else:
                    d = (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
                    print("%-12s max diff vs first variant: %.2e" % (name, d))
    for name, (co, to) in times.items():
        print("%-12s core median %.3f ms (min %.3f)   whole median %.3f ms (min %.3f)" % (name, statistics.median(co), min(co), statistics.median(to), min(to)))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 5, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
