"""One shape of the plain-bf16 projection GEMM, a few launches (rocprofv3 --pmc target): python tools/pmc_gemm16_one.py M N K [stream_k]"""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
ops = cti_amd.pkg.ops
M, N, K = (int(x) for x in sys.argv[1:4])
sk = len(sys.argv) > 4 and sys.argv[4] == "1"
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16); w = (torch.randn(N, K, generator=g) / 8).cuda(); b = torch.randn(N, generator=g).cuda()
wp = ops.split_operand(w, prec="bf16")
for _ in range(6):
    y = ops.gemm_bf16_rows(a, wp, N, out_dtype=torch.bfloat16, bias=b, relu=True, stream_k=sk)
torch.cuda.synchronize()
