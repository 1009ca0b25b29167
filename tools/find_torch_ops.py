"""Which PyTorch (aten) kernels still run inside one eval forward of the configs[2] / [3] models, and from which lines of this package?  (round 5; GPU box)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
import cti_amd

cti_amd.set_precision("bf16")


VIEW_OPS = {"view", "_unsafe_view", "slice", "select", "detach", "alias", "expand", "as_strided", "t", "transpose", "unsqueeze", "squeeze", "reshape", "permute",
            "empty", "empty_like", "empty_strided", "new_empty", "sym_size", "sym_stride", "is_pinned", "_local_scalar_dense", "record_stream", "unbind", "split"}


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        base = name.split(".")[1] if name.count(".") >= 1 else name
        if base not in VIEW_OPS:
            fr = [x for x in traceback.extract_stack()[:-1] if "iccv19" in x.filename or "cti_amd" in x.filename][-2:]
            self.seen[(name, " <- ".join("%s:%d %s" % (os.path.basename(x.filename), x.lineno, x.name) for x in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))


for cfg in ("c3", "c4"):
    S = bench.model_setup(cfg, 256, 0, torch.device("cuda:0"))
    os.environ["CTI_BENCH_SERIAL_MODELS"] = "1"
    with torch.no_grad():
        for _ in range(3):
            S["fwd"]()
        torch.cuda.synchronize()
        with Spy() as spy:
            S["fwd"]()
        torch.cuda.synchronize()
    print("==", cfg)
    for k, c in spy.seen.most_common():
        print("  %2d x %-32s %s" % (c, k[0], k[1]))
