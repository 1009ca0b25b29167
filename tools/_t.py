import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import cti_amd
ops = cti_amd.pkg.ops
cti_amd.set_precision("bf16")
g = torch.Generator().manual_seed(0)
res = {}
for M, n, N, K in [(3072, 2, 1024, 1024), (1536, 2, 1024, 1024), (3584, 2, 1024, 1024), (2304, 1, 512, 512), (9216, 1, 512, 512), (768, 1, 512, 512), (3584, 8, 1024, 1024)]:
    a = torch.randn(M, K, generator=g).cuda(); w = torch.randn(n * N, K, generator=g).cuda() / K ** 0.5
    bias = torch.randn(n * N, generator=g).cuda()
    wp = ops.split_operand(w)
    f = lambda: ops.gemm_nt(a, w, nb1=n, rA1=0, rB1=N, M=M, N=N, bias=bias, bias_bs=N, B_planes=wp)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): f()
        gr = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(20): o = f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    res["%dx(%dx%d)x%d" % (M, n, N, K)] = round(sorted(ts)[2], 1)
print(json.dumps({"rs": os.environ.get("CTI_GEMM_RS", "1"), "small16": os.environ.get("CTI_GEMM16_SMALL", "1"), "graph replay, us per (split + product)": res}))
