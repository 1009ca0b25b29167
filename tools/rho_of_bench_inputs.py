import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, cti_amd
import bench
ops = cti_amd.ops
ops._range_debug = True
c = dict(bench.C2)
torch.manual_seed(bench.SEED)
net = cti_amd.TCNet(c["v_dim"], c["q_dim"], c["a_dim"], c["h_mm"], 1, c["rank"], c["glimpse"]).cuda().eval()
for seed in (bench.SEED + 1, 7, 8):
    v, q, a = bench.synth_inputs(c, c["B"], seed, torch.device("cuda"))
    with torch.no_grad():
        out = net(v, q, a)
    torch.cuda.synchronize()
    print("bench config seed", seed, ops.f16f6_range_status())
    del out
