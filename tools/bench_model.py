#!/usr/bin/env python3
"""Model-level timings at the reference's real widths (rows N1, N3, N4 of 8f): full forward (eval) and
one training step (forward + loss + backward + clipped Adamax through FlatAdamaxDP) of the FFOE CTI model, the FFOE BAN model
(gamma = 8) and the MC CTI model, B = 256, random-init weights, synthetic inputs.  One JSON line per case.

    python tools/bench_model.py [ffoe_cti ffoe_ban mc_cti] [--train] [--steps 20] [--precision bf16x3|bf16|fp32]

`--precision bf16` = plain bf16 products with fp32 accumulation (BASELINE configs[2], [3] name bf16): 1 MFMA per product instead of 3;
its own tolerance is in tests/test_parity_gpu.py::test_plain_bf16_mode_within_its_own_tolerance.
"""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd  # noqa: E402

DEV = "cuda"


class DS:
    def __init__(self, ntoken, v_dim, num_ans):
        self.dictionary = types.SimpleNamespace(ntoken=ntoken)
        self.v_dim = v_dim
        self.num_ans_candidates = num_ans


def args_of(gamma):
    return types.SimpleNamespace(op="c", num_hid=1024, gamma=gamma, h_mm=512, rank=32, k=1, h_out=1, activation="relu", dropout=0.5,
                                 use_counter=False)


def tokens(B, L, ntoken, g):
    t = torch.randint(0, ntoken, (B, L), generator=g)
    n = torch.randint(3, L + 1, (B,), generator=g)
    t[torch.arange(L)[None, :] >= n[:, None]] = ntoken
    return t


CASES = {
    # name: (builder, gamma, num_ans, Q, A, forward signature)
    "ffoe_cti": ("build_cti", 2, 3129, 14, 3),
    "ffoe_ban": ("build_ban", 8, 3129, 14, 0),
    "mc_cti": ("build_mc_cti", 2, 2, 12, 6),
    "mc_ban": ("build_mc_ban", 2, 2, 12, 6),
}


def run(name, train, steps, warmup=5, B=256, ntoken=20000, around_timed=None):
    builder, gamma, num_ans, Q, A = CASES[name]
    torch.manual_seed(1204)
    m = getattr(cti_amd, builder)(args_of(gamma), DS(ntoken, 2048, num_ans)).to(DEV)
    g = torch.Generator().manual_seed(7)
    v = torch.randn(B, 36, 2048, generator=g).abs()
    nv = torch.randint(10, 37, (B,), generator=g)
    v[torch.arange(36)[None, :] >= nv[:, None]] = 0
    v = v.to(DEV)
    q = tokens(B, Q, ntoken, g).to(DEV)
    a = tokens(B, A, ntoken, g).to(DEV) if A else None
    boxes = torch.rand(B, 36, 6, generator=g).to(DEV)
    tgt = ((torch.rand(B, num_ans, generator=g) < 0.01).float() * torch.rand(B, num_ans, generator=g)).to(DEV)

    def fwd():
        if name == "ffoe_cti":
            return m(v, q, a)
        if name == "ffoe_ban":
            return m(v, boxes, q, None)[0]
        if name == "mc_ban":
            return m(v, boxes, q, a)[0]
        return m(v, boxes, q, a)[0]

    if train:
        m.train()
        opt = cti_amd.FlatAdamaxDP(m, lr=1e-3, clip_norm=0.25)
        crit = cti_amd.BCEWithLogitsSum()

        def step():
            opt.zero_grad()
            loss = crit(fwd(), tgt) / B
            loss.backward()
            opt.step()
    else:
        m.eval()

        def step():
            with torch.no_grad():
                fwd()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if around_timed:
        around_timed[0]()                                     # e.g. a host profiler around exactly the timed loop (tools/cpu_profile_train.py)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0                         # launch-side time: the GPU may still be running
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    if around_timed:
        around_timed[1](t_host / steps * 1e3)
    nparam = sum(p.numel() for p in m.parameters())
    print(json.dumps({"case": name, "mode": "train_step" if train else "forward", "B": B, "ms": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1),
                      "params": nparam, "precision": cti_amd.get_precision(), "Q": Q, "A": A, "gamma": gamma}), flush=True)


if __name__ == "__main__":
    names = [x for x in sys.argv[1:] if x in CASES] or list(CASES)
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    if "--precision" in sys.argv:
        cti_amd.set_precision(sys.argv[sys.argv.index("--precision") + 1])
    for n in names:
        run(n, "--train" in sys.argv, steps)
