import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cti_amd
ops = cti_amd.ops
g = torch.Generator().manual_seed(0)
a = torch.randn(256 * 3129, 300, generator=g).cuda()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: ops.quantize_f16f6(a, 0))
print(json.dumps({"quantize_a_ms_incl_memset": round(ms, 4), "qrows": os.environ.get("CTI_F6_QROWS", "1")}))
