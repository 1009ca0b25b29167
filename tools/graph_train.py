#!/usr/bin/env python3
"""Training step of the FFOE CTI model (B = 256, the reference's widths) eager vs captured in a hipGraph (cti_amd.GraphedTrainStep):
ms per step and the host-side share.   python tools/graph_train.py [ffoe_cti|ffoe_ban|mc_cti] [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cti_amd  # noqa: E402
import bench_model as bm  # noqa: E402


def main(name, steps):
    builder, gamma, num_ans, Q, A = bm.CASES[name]
    B, ntoken = 256, 20000
    torch.manual_seed(1204)
    m = getattr(cti_amd, builder)(bm.args_of(gamma), bm.DS(ntoken, 2048, num_ans)).to("cuda").train()
    g = torch.Generator().manual_seed(7)
    v = torch.randn(B, 36, 2048, generator=g).abs().cuda()
    q = bm.tokens(B, Q, ntoken, g).cuda()
    a = bm.tokens(B, A, ntoken, g).cuda() if A else None
    boxes = torch.rand(B, 36, 6, generator=g).cuda()
    tgt = ((torch.rand(B, num_ans, generator=g) < 0.01).float()).cuda()
    crit = cti_amd.BCEWithLogitsSum()
    opt = cti_amd.FlatAdamaxDP(m, lr=1e-3, clip_norm=0.25)
    if name == "ffoe_cti":
        inputs, fwd = (v, q, a), (lambda v_, q_, a_: m(v_, q_, a_))
    elif name == "ffoe_ban":
        inputs, fwd = (v, boxes, q), (lambda v_, b_, q_: m(v_, b_, q_, None)[0])
    else:
        inputs, fwd = (v, boxes, q, a), (lambda v_, b_, q_, a_: m(v_, b_, q_, a_)[0])

    class Wrap(torch.nn.Module):
        def forward(self, *xs):
            return fwd(*xs)
    loss_fn = lambda out, t: crit(out, t) / B        # noqa: E731

    def eager():
        opt.zero_grad()
        loss_fn(fwd(*inputs), tgt).backward()
        opt.step()

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, th / steps * 1e3
    t_e, h_e = timeit(eager)
    gs = cti_amd.GraphedTrainStep(Wrap(), opt, loss_fn, inputs, tgt, warmup=2)
    t_g, h_g = timeit(lambda: gs(inputs, tgt))
    print(json.dumps({"case": name, "B": B, "eager_ms": round(t_e, 3), "eager_host_ms": round(h_e, 3), "graph_ms": round(t_g, 3), "graph_host_ms": round(h_g, 3),
                      "samples_per_s_graph": round(B / t_g * 1e3, 1), "precision": cti_amd.get_precision(), "steps_done": opt.steps_done()}))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "ffoe_cti", int(sys.argv[2]) if len(sys.argv) > 2 else 30)
