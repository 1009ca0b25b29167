"""Where do the device-to-device copies of a model forward come from?  (torch.profiler with stacks; run on the GPU box:
   python tools/find_copies.py c3)  Prints every aten::copy_ / clone / contiguous / cat of one eager forward with the innermost package frame."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
import cti_amd
cti_amd.set_precision("bf16")
s = bench.model_setup(cfg, 256, 0, dev)
with torch.no_grad():
    for _ in range(3):
        s["fwd"]()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        s["fwd"]()
        torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::repeat_interleave"):
        fr = [f for f in (e.stack or []) if "iccv19" in f or "bench.py" in f]
        cnt[(e.name, str(e.input_shapes)[:80], " <- ".join(x.split("/")[-1] for x in fr[:3]))] += 1
for k, n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(n, *k)
