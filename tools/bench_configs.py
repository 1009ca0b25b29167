#!/usr/bin/env python3
"""Timings of the hot-path modules at the other BASELINE / SURVEY configurations (C1, C3, C4 shapes) on one MI355X, next to
the numpy oracle on the host cores.  These are parity-test shapes, not the bench metric; the numbers go into DESIGN.md."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cti_amd                                                     # noqa: E402
from oracle import cti_oracle as O                                 # noqa: E402

DEV = "cuda"


def gpu_ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def cpu_ms(fn, budget=3.0):
    fn()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < budget:
        fn(); n += 1
    return (time.perf_counter() - t0) / n * 1e3


def sd(m):
    return {k: t.detach().cpu().numpy() for k, t in m.state_dict().items()}


def main():
    torch.manual_seed(1204)
    out = {}
    g = torch.Generator().manual_seed(7)
    rnd = lambda *s: torch.randn(*s, generator=g)
    # C1: TCNet.forward B=4, V=36x2048, Q=14x600, A=4x300
    m = cti_amd.TCNet(2048, 600, 300, 512, 1, 32, 2).to(DEV).eval()
    v, q, a = rnd(4, 36, 2048).abs(), rnd(4, 14, 600), rnd(4, 4, 300)
    vd, qd, ad = v.to(DEV), q.to(DEV), a.to(DEV)
    with torch.no_grad():
        out["C1 TCNet.forward B=4 (ms)"] = {"gpu": gpu_ms(lambda: m(vd, qd, ad)), "oracle_cpu": cpu_ms(lambda: O.tcnet_forward(v.numpy(), q.numpy(), a.numpy(), sd(m)))}
    # C4 shapes: TriAttention B=256, V=36x2048, Q=12x1024, A=3x1024; t_net fww; BiAttention G=8; b_net fww
    B = 256
    v, q, a = rnd(B, 36, 2048).abs(), torch.tanh(rnd(B, 12, 1024)), torch.tanh(rnd(B, 3, 1024))
    vd, qd, ad = v.to(DEV), q.to(DEV), a.to(DEV)
    tri = cti_amd.TriAttention(2048, 1024, 1024, 512, 1, 32, 2, 1).to(DEV).eval()
    tnet = cti_amd.TCNet(2048, 1024, 1024, 512, 1, 32, 1, k=2).to(DEV).eval()
    bi = cti_amd.BiAttention(2048, 1024, 1024, 8).to(DEV).eval()
    bnet = cti_amd.BCNet(2048, 1024, 1024, None, k=1).to(DEV).eval()
    with torch.no_grad():
        p, _ = tri(vd, qd, ad)
        pb, _ = bi.forward_all(vd, qd)
        out["C4 TriAttention B=256 (ms)"] = {"gpu": gpu_ms(lambda: tri(vd, qd, ad))}
        out["C4 TCNet.forward_with_weights B=256 (ms)"] = {"gpu": gpu_ms(lambda: tnet.forward_with_weights(vd, qd, ad, p[..., 0]))}
        out["C4 BiAttention G=8 B=256 (ms)"] = {"gpu": gpu_ms(lambda: bi.forward_all(vd, qd))}
        out["C4 BCNet.forward_with_weights B=256 (ms)"] = {"gpu": gpu_ms(lambda: bnet.forward_with_weights(vd, qd, pb[:, 0]))}
    Bc = 16
    vs, qs, as_ = v[:Bc].numpy(), q[:Bc].numpy(), a[:Bc].numpy()
    out["C4 TriAttention B=256 (ms)"]["oracle_cpu_scaled_from_B16"] = cpu_ms(lambda: O.tri_attention(vs, qs, as_, sd(tri))) * B / Bc
    out["C4 BiAttention G=8 B=256 (ms)"]["oracle_cpu_scaled_from_B16"] = cpu_ms(lambda: O.bi_attention(vs, qs, sd(bi))) * B / Bc
    # one training step of the C4-shaped CTI fusion block (forward + backward through HIP kernels)
    tri.train(); tnet.train()
    qd.requires_grad_(True); ad.requires_grad_(True)

    def train_step():
        for prm in list(tri.parameters()) + list(tnet.parameters()):
            prm.grad = None
        p_, _ = tri(vd, qd, ad)
        o = tnet.forward_with_weights(vd, qd, ad, p_[..., 0])
        o.sum().backward()
    out["C4 TriAttention + t_net fwd+bwd (train mode) B=256 (ms)"] = {"gpu": gpu_ms(train_step, reps=5, warm=2)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
