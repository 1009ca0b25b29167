#!/usr/bin/env python3
"""Host-side cost of one training step: cProfile around exactly the timed loop of tools/bench_model.py (backward runs in the calling thread
so its frames are seen).  Prints the host time per step next to the wall time per step: when they are close, the step is launch-bound.
python tools/cpu_profile_train.py [ffoe_cti] [steps]"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_model as bm
import torch
name = sys.argv[1] if len(sys.argv) > 1 else "ffoe_cti"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
host = []
bm.run(name, True, steps, around_timed=(lambda: None, lambda h: host.append(h)))          # unprofiled: host ms / step vs wall ms / step
print("host ms/step (no profiler): %.3f" % host[0])
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
bm.run(name, True, steps, around_timed=(pr.enable, lambda h: pr.disable()))
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22)
    print(s.getvalue()[:4500])
