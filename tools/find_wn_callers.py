"""Which call sites still compute a weight-norm scale on every eval forward of the configs[2] / [3] models?  (round 5; run on the GPU box)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import cti_amd
from cti_amd import ops

cti_amd.set_precision("bf16")
for cfg in ("c3", "c4"):
    S = bench.model_setup(cfg, 256, 0, torch.device("cuda:0"))
    with torch.no_grad():
        for _ in range(3):
            S["fwd"]()
        torch.cuda.synchronize()
        seen = collections.Counter()
        real = {n: getattr(ops, n) for n in ("wn_scale", "wn_scale_many", "split_operand", "zero_fill") if hasattr(ops, n)}
        def wrap(n, f):
            def g(*a, **k):
                fr = [x for x in traceback.extract_stack()[:-1] if "cti_amd" in x.filename or "iccv19" in x.filename][-2:]
                seen[(n, " <- ".join("%s:%d %s" % (os.path.basename(x.filename), x.lineno, x.name) for x in reversed(fr)))] += 1
                return f(*a, **k)
            return g
        for n, f in real.items():
            setattr(ops, n, wrap(n, f))
        S["fwd"]()
        torch.cuda.synchronize()
        for n, f in real.items():
            setattr(ops, n, f)
    print("==", cfg)
    for k, c in seen.most_common():
        print("  %2d x %s: %s" % (c, k[0], k[1]))
