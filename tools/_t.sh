for lib in "" tpm_abl1 tpm_abl2; do
echo "== variant: ${lib:-shipped}"
if [ -n "$lib" ]; then export CTI_HIP_LIB=$GRAFT_REPO_ROOT/iccv19_vqa-cti_amd/lib/variants/libcti_hip_$lib.so; fi
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import importlib.util
spec = importlib.util.spec_from_file_location("bp", "tools/bench_pools.py"); bp = importlib.util.module_from_spec(spec); spec.loader.exec_module(bp)
from cti_amd import ops
torch.manual_seed(0)
B, V, D = 256, 36, 1024
for (Q, A) in ((14, 3), (12, 6)):
    vt = torch.randn(B, V, D, device="cuda"); qt = torch.randn(B, Q, D, device="cuda"); at = torch.randn(B, A, D, device="cuda")
    att = torch.softmax(torch.randn(B, V * Q * A, 2, device="cuda"), 1).view(B, V, Q, A, 2)
    for name, w in (("strided", att[..., 0]), ("contiguous", att[..., 0].contiguous())):
        ts = sorted(bp.timeit(lambda: ops.tri_pool(vt, qt, at, w), 50) for _ in range(5))
        print("A=%d w %s median %.2f us  min %.2f" % (A, name, ts[2], ts[0]))
PY
done
