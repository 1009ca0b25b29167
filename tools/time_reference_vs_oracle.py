#!/usr/bin/env python3
"""Build-container check: the GENUINE reference TCNet.forward (imported from /root/reference) timed beside
the numpy oracle at the BASELINE configs[1] per-sample shapes on the same inputs and cores, so that the GPU box's `cpu_baseline` (the oracle --
the reference's Python cannot travel) can be read as a stand-in for the reference.  Run here only:
    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tools/time_reference_vs_oracle.py
The process puts /root/repo/oracle and /root/repo/tests on the path by file location (no top-level `src` collision: the reference's `src`
package comes from the cwd)."""
import importlib.util
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
assert os.path.isfile("src/tc.py"), "run with cwd=/root/reference"
sys.path.insert(0, os.getcwd())
from src.tc import TCNet  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


O = load("cti_oracle", os.path.join(REPO, "oracle", "cti_oracle.py"))
threads = int(os.environ.get("THREADS", os.cpu_count() or 1))
torch.set_num_threads(threads)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(1204)
net = TCNet(2048, 1024, 300, 512, 1, 32, 2).eval()
g = torch.Generator().manual_seed(1)
v = torch.randn(B, 36, 2048, generator=g).abs()
q = torch.randn(B, 14, 1024, generator=g)
a = torch.randn(B, 3129, 300, generator=g)
state = {k: t.detach().numpy() for k, t in net.state_dict().items()}


def med(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


with torch.no_grad():
    ref = net(v, q, a).contiguous().numpy()
    t_ref = med(lambda: net(v, q, a))
orc = O.tcnet_forward(v.numpy(), q.numpy(), a.numpy(), state)
t_orc = med(lambda: O.tcnet_forward(v.numpy(), q.numpy(), a.numpy(), state))
err = float(np.max(np.abs(orc - ref)) / np.max(np.abs(ref)))
print(json.dumps({"shape": "BASELINE configs[1] per-sample shapes, B=%d" % B, "torch_threads": threads, "cpu_count": os.cpu_count(),
                  "reference_ms": round(t_ref * 1e3, 1), "reference_samples_per_s": round(B / t_ref, 2),
                  "oracle_ms": round(t_orc * 1e3, 1), "oracle_samples_per_s": round(B / t_orc, 2),
                  "oracle_vs_reference_norm_max_err": err}))
