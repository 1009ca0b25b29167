#!/usr/bin/env python3
"""hipGraph capture of a whole eval forward (torch.cuda.CUDAGraph around the model call): the library allocates nothing and never
synchronises, so the launch sequence -- including the two-stream section of cti_tcnet_forward -- is capturable.  Prints eager vs
replay time and the max difference of the outputs.   python tools/graph_model.py [ffoe_cti|ffoe_ban|mc_cti]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cti_amd  # noqa: E402
import bench_model as bm  # noqa: E402


def main(name):
    builder, gamma, num_ans, Q, A = bm.CASES[name]
    B, ntoken = 256, 20000
    torch.manual_seed(1204)
    m = getattr(cti_amd, builder)(bm.args_of(gamma), bm.DS(ntoken, 2048, num_ans)).to("cuda").eval()
    g = torch.Generator().manual_seed(7)
    v = torch.randn(B, 36, 2048, generator=g).abs().cuda()
    q = bm.tokens(B, Q, ntoken, g).cuda()
    a = bm.tokens(B, A, ntoken, g).cuda() if A else None
    boxes = torch.rand(B, 36, 6, generator=g).cuda()

    def fwd():
        with torch.no_grad():
            if name == "ffoe_cti":
                return m(v, q, a)
            if name == "ffoe_ban":
                return m(v, boxes, q, None)[0]
            return m(v, boxes, q, a)[0]

    for _ in range(3):
        ref = fwd()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=s):
            out = fwd()
    torch.cuda.synchronize()

    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_eager = timeit(fwd)
    t_graph = timeit(graph.replay)
    graph.replay(); torch.cuda.synchronize()
    print("%s: eager %.3f ms, graph replay %.3f ms, max |diff| %.3g (ref max %.3g)" % (name, t_eager, t_graph, float((out - ref).abs().max()), float(ref.abs().max())))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "ffoe_cti")
