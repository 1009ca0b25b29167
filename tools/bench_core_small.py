"""The fused few-answer TriAttention (reference src/attention.py:49-59 on src/tc.py:41-52 with A <= 6) at the model shapes, 30 calls (rocprofv3 --kernel-trace --stats target)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd
cti_amd.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
torch.manual_seed(0)
for (B, V, Q, A) in ((256, 36, 12, 6), (256, 36, 14, 3)):
    m = cti_amd.TriAttention(2048, 1024, 1024, 512, 1, 32, 2, 1).cuda().eval()
    v = torch.randn(B, V, 2048, device="cuda").abs(); q = torch.randn(B, Q, 1024, device="cuda"); a = torch.randn(B, A, 1024, device="cuda")
    with torch.no_grad():
        for _ in range(30):
            p, lg = m(v, q, a)
    torch.cuda.synchronize()
    print(B, V, Q, A, float(p.sum()))
