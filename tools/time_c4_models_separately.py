#!/usr/bin/env python3
"""configs[3]'s two models, each ALONE, as a replayed hipGraph (bf16 mode, B = 256): what the sibling-stream forward (bench.py --config c4) has to beat."""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cti_amd, bench
cti_amd.set_precision("bf16")
dev = torch.device("cuda")
s = bench.model_setup("c4", 256, 0, dev)
ban, cti = s["models"]["ban"], s["models"]["cti"]
# the batch bench.model_setup drew lives in the closure of s['fwd']: draw the same shapes again
g = torch.Generator().manual_seed(5)
v = torch.randn(256, 36, 2048, generator=g).abs().to(dev)
q = torch.randint(0, 20000, (256, 14), generator=g).to(dev)
a = torch.randint(0, 20000, (256, 3), generator=g).to(dev)
res = {}
with torch.no_grad():
    for name, fn in (("ban", lambda: ban(v, None, q, None)[0]), ("cti", lambda: cti(v, q, a)),
                     ("both_serial", lambda: (ban(v, None, q, None)[0], cti(v, q, a))),
                     ("both_sibling_streams", lambda: cti_amd.ops.run_concurrently(lambda: cti(v, q, a), lambda: ban(v, None, q, None)[0]))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr, stream=st):
                fn()
        torch.cuda.current_stream().wait_stream(st)
        for _ in range(5): gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): gr.replay()
        e1.record(); torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / 50, 4)
print(json.dumps({"ms_per_forward_graph_replay_B256_bf16": res}))
