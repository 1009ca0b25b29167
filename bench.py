#!/usr/bin/env python3
"""bench.py -- CTI fused-forward samples/sec (BASELINE.json metric) on N MI355X GPUs of one node.

One "step" = one TCNet.forward (reference src/tc.py:41-52) over one batch of synthetic tensors of BASELINE.json
configs[1]: B=256 per GPU, V=36x2048, Q=14x1024, A=3129x300, rank=32, h_mm=512, glimpse=2, fp32 in / fp32 out,
inputs resident in HBM before the timed region.  The batch axis shards across GPUs with no data-path collective
(forward-only replicas, weak scaling: 256 rows per GPU); the only collectives are the timing barrier and the
max-over-ranks of the elapsed time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp32|bf16x3] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the definition of every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs[1]
C2 = dict(B=256, V=36, Q=14, A=3129, v_dim=2048, q_dim=1024, a_dim=300, h_mm=512, rank=32, glimpse=2)
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0}   # MI355X_MICROARCH.md: f32 MFMA / dense bf16 MFMA
SEED = 1204                                                         # the reference's default seed (src/FFOE/main.py:53)


def flops_per_sample(c):
    """tucker = 2h(V*vd + Q*qd + A*ad), rank = 2h^2(V+Q+A), core = mode-1/2 + 2*V*Q*A*G*h."""
    h, hr, R, G = c["h_mm"], c["h_mm"] // c["rank"], c["rank"], c["glimpse"]
    tucker = 2 * h * (c["V"] * c["v_dim"] + c["Q"] * c["q_dim"] + c["A"] * c["a_dim"])
    rank = 2 * h * h * (c["V"] + c["Q"] + c["A"])
    core_final = 2 * c["V"] * c["Q"] * c["A"] * G * h
    core_12 = 2 * c["V"] * R * hr * hr * G * hr + 2 * c["V"] * c["Q"] * R * hr * G * hr
    return dict(tucker=tucker, rank=rank, core_final=core_final, core_12=core_12, total=tucker + rank + core_final + core_12)


def synth_inputs(c, B, seed, device):
    """v ~ |N(0,1)| with a random number of trailing all-zero rows per sample (bottom-up features are post-ReLU and
    zero-padded by trim_collate, reference src/utils.py:127-136); q, a ~ N(0,1) at the raw widths configs[1] names."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    v = torch.randn(B, c["V"], c["v_dim"], generator=g).abs_()
    nv = torch.randint(10, c["V"] + 1, (B,), generator=g)
    for b in range(B):
        v[b, int(nv[b]):] = 0
    q = torch.randn(B, c["Q"], c["q_dim"], generator=g)
    a = torch.randn(B, c["A"], c["a_dim"], generator=g)
    return v.to(device), q.to(device), a.to(device)


def cpu_baseline(c, state, seed, budget_s=20.0):
    """The oracle (numpy restatement of the reference's CPU path, oracle/cti_oracle.py) timed on this host's cores on a
    bounded sample of the same workload: B_cpu samples of the C2 shapes, repeated until ~budget_s of CPU work."""
    from oracle import cti_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    Bc = 4
    v, q, a = synth_inputs(c, Bc, seed + 17, "cpu")
    v, q, a = v.numpy(), q.numpy(), a.numpy()
    O.tcnet_forward(v[:1], q[:1], a[:1], state)                  # warm-up (BLAS threads, page faults)
    t0 = time.perf_counter()
    n = 0
    while True:
        O.tcnet_forward(v, q, a, state)
        n += Bc
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 1024:
            break
    return {"value": n / el, "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": "oracle.tcnet_forward (numpy fp32), %d samples of the C2 shapes in batches of %d, %.1f s" % (n, Bc, el)}


def measure(step, steps, warmup, world, sync, dist=None, device="cpu"):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both sides; returns the
    MAX over ranks of the elapsed seconds (the contract's timing rule).  `sync` = torch.cuda.synchronize on a GPU."""
    def barrier():
        if dist is not None and dist.is_initialized():
            dist.barrier()
    for _ in range(warmup):
        step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync(); barrier(); sync()
    el = time.perf_counter() - t0
    t = torch.tensor([el], device=device, dtype=torch.float64)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    flush_c_stdio()
    return el


def flush_c_stdio():
    """RCCL prints a version banner through C stdio when the first communicator is created; on a pipe it would sit in libc's buffer until
    exit and land AFTER the JSON line.  Push it out now so that the JSON line is the last line of stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def whole_job_rate(world, batch_per_rank, steps, elapsed):
    """samples/s of the whole job: every rank processed batch_per_rank * steps samples in `elapsed` (max over ranks)."""
    return world * batch_per_rank * steps / elapsed


class CTIFusionBlock(torch.nn.Module):
    """The CTI fusion of CTIModel.forward (reference src/FFOE/base_model.py:128-135) at the VQA-2.0 shapes of BASELINE configs[3]/[4]:
    TriAttention (h_mm 512, rank 32, glimpse 2), one TCNet(k=2) pooling + q_prj/a_prj residual per glimpse, sum, and a stock-torch
    2-layer classifier (outside the CTI path).  Used by --mode train only."""

    def __init__(self, cti, v_dim=2048, num_hid=1024, h_mm=512, rank=32, gamma=2, n_ans=3129):
        super().__init__()
        self.gamma = gamma
        self.t_att = cti.TriAttention(v_dim, num_hid, num_hid, h_mm, 1, rank, gamma, 1, dropout=[.2, .5])
        self.t_net = torch.nn.ModuleList([cti.TCNet(v_dim, num_hid, num_hid, h_mm, 1, rank, 1, dropout=[.2, .5], k=2) for _ in range(gamma)])
        self.q_prj = torch.nn.ModuleList([cti.FCNet([num_hid, num_hid], '', .2) for _ in range(gamma)])
        self.a_prj = torch.nn.ModuleList([cti.FCNet([num_hid, num_hid], '', .2) for _ in range(gamma)])
        self.classifier = torch.nn.Sequential(torch.nn.Linear(num_hid, 2 * num_hid), torch.nn.ReLU(), torch.nn.Linear(2 * num_hid, n_ans))

    def forward(self, v, q_emb, ans_emb):
        att, _ = self.t_att(v, q_emb, ans_emb)
        for g in range(self.gamma):
            b_emb = self.t_net[g].forward_with_weights(v, q_emb, ans_emb, att[:, :, :, :, g])
            q_emb = self.q_prj[g](b_emb.unsqueeze(1)) + q_emb
            ans_emb = self.a_prj[g](b_emb.unsqueeze(1)) + ans_emb
        return self.classifier(q_emb.sum(1) + ans_emb.sum(1))


def run_train(args, world, rank, dev, dist):
    """--mode train: data-parallel training step (BASELINE configs[4] shape: 256 rows per GPU) of the CTI fusion block:
    forward + backward in HIP, ONE RCCL all-reduce of the flat gradient buffer, fused clip + Adamax.  Not the headline metric."""
    import cti_amd
    cti_amd.set_precision(args.precision)
    torch.manual_seed(SEED)
    model = CTIFusionBlock(cti_amd).to(dev).train()
    opt = cti_amd.FlatAdamaxDP(model, lr=1e-3, clip_norm=0.25)
    opt.broadcast_parameters()
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(SEED + 1 + rank)
    v = torch.randn(B, 36, 2048, generator=g).abs_()
    for b in range(B):
        v[b, int(torch.randint(10, 37, (1,), generator=g)):] = 0
    q = torch.tanh(torch.randn(B, 12, 1024, generator=g)).to(dev)
    a = torch.tanh(torch.randn(B, 3, 1024, generator=g)).to(dev)
    y = (torch.rand(B, 3129, generator=g) > 0.999).float().to(dev)
    v = v.to(dev)

    def step():
        opt.zero_grad()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(model(v, q, a), y, reduction="sum") / B
        loss.backward()
        opt.step()

    el = measure(step, args.steps, args.warmup, world, torch.cuda.synchronize, dist, dev)
    if dist.is_initialized():                                   # tear RCCL down first: the JSON line must be the last line of stdout
        dist.barrier(); dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps({"metric": "CTI fusion-block data-parallel training samples/sec (256 rows/GPU, VQA-2.0 shapes)",
                          "value": whole_job_rate(world, B, args.steps, el), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32 (bf16x3 split products)" if args.precision == "bf16x3" else args.precision,
                          "data": "synthetic",
                          "config": {"workload": "BASELINE configs[4] shape: TriAttention + 2 x (TCNet.forward_with_weights, q_prj, a_prj) + classifier, "
                                                 "train mode (dropout on), fwd + bwd + one all-reduce + fused clip/Adamax",
                                     "global_batch": world * B, "parameters": opt.n_params, "parallelism": "dp%d" % world}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="forward", choices=["forward", "train"], help="forward (the BASELINE metric) or train (DP step)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (the default run takes well under a minute on the GPU + ~20 s of CPU baseline)")
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps: allocator growth, one-time kernel attributes, clock ramp")
    ap.add_argument("--batch", type=int, default=C2["B"], help="rows per GPU (default 256 = BASELINE configs[1])")
    ap.add_argument("--precision", default=os.environ.get("CTI_PRECISION", "bf16x3"), choices=["fp32", "bf16x3", "bf16"],
                    help="bf16x3 (default): 3-term split-bf16 MFMA, fp32-grade (1e-5 vs the float64 oracle); fp32: exact fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    import torch.distributed as dist
    # CTI_BENCH_FORCE_DIST=1: initialise RCCL even for one rank (exercises the N > 1 code path -- init, barrier, max-reduce -- on a 1-GPU box)
    if world > 1 or os.environ.get("CTI_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.mode == "train":
        run_train(args, world, rank, dev, dist)
        if dist is not None and dist.is_initialized():
            dist.barrier(); dist.destroy_process_group()
        return
    import cti_amd
    cti_amd.set_precision(args.precision)
    c = dict(C2, B=args.batch)
    torch.manual_seed(SEED)                                      # identical parameters on every rank
    net = cti_amd.TCNet(c["v_dim"], c["q_dim"], c["a_dim"], c["h_mm"], 1, c["rank"], c["glimpse"]).to(dev).eval()
    v, q, a = synth_inputs(c, c["B"], SEED + 1 + rank, dev)      # a different shard of the global batch per rank

    res_holder = {}

    def step():
        res_holder["out"] = net(v, q, a)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        cti_amd.ops.profile_start()                            # hipEvents around the kernels of the timed steps
        el = measure(step, args.steps, 0, world, torch.cuda.synchronize, dist, dev)
        kt = cti_amd.ops.profile_stop()
    out = res_holder["out"]
    assert out.shape == (c["B"], c["V"], c["Q"], c["A"], c["glimpse"]) and bool(torch.isfinite(out[0, 0, 0, 0]).all())
    if rank == 0:
        fl = flops_per_sample(c)
        core_ms = float(np.mean(kt.get("paralind_core", kt.get("tcnet_forward"))))
        core_flops = fl["core_final"] * c["B"]                   # algorithmic flops of ONE launch of the dominant kernel
        achieved = core_flops / (core_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        kern = {k: round(float(np.mean(ms)), 3) for k, ms in sorted(kt.items())}
        # HBM bytes of the dominant kernel come from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled
        # as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is); the summary is committed under profiles/.
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "core_traffic.json")
        if args.precision == "bf16x3" and c["B"] == C2["B"] and os.path.isfile(tf):
            tj = json.load(open(tf))
            traffic, traffic_src = tj["hbm_bytes_per_launch"], tj["source"]
        mfma_per_flop = 3.0 if args.precision == "bf16x3" else 1.0
        if args.precision == "bf16":
            res_note = "plain-bf16 mode is NOT the BASELINE metric (configs[1] is fp32): reported for reference only"
        res = {
            "metric": "CTI fused-forward samples/sec at B=256 (V=36x2048)",
            "value": whole_job_rate(world, c["B"], args.steps, el), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "f32 (bf16x3 split products, f32 accumulate)", "bf16": "bf16 products, f32 accumulate"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: TCNet.forward fp32, B=%d/GPU, V=36x2048, Q=14x1024, A=3129x300, rank=32, "
                                   "h_mm=512, glimpse=2" % c["B"], "global_batch": world * c["B"], "precision": args.precision,
                       "parallelism": "replicas x%d (batch-sharded, no data-path collective)" % world,
                       "gflop_per_sample": round(fl["total"] / 1e9, 4)},
            "roofline": {"bound": "mfma", "kernel": "paralind_core (mode-3 product + rank sum, batched NT GEMM 504x3129x512 x%d)" % (c["B"] * c["glimpse"]),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "launch_ms": core_ms, "flops_per_launch": core_flops,
                         "mfma_issued_tflops": achieved * mfma_per_flop, "mfma_issued_frac": achieved * mfma_per_flop / peak,
                         "algorithmic_bytes_per_launch": c["B"] * 4 * (c["V"] * c["Q"] * c["A"] * c["glimpse"] + c["h_mm"] * (c["V"] * c["Q"] * c["glimpse"] + c["A"])),
                         "traffic_source": traffic_src,
                         "note": "achieved = algorithmic fp32 flops / launch time; bf16x3 issues 3 bf16 MFMAs per product, so the MFMA pipes "
                                 "run at mfma_issued_tflops against the 2500 TFLOP/s dense bf16 peak"},
            "whole_step_tflops": fl["total"] * c["B"] * args.steps / el / 1e12,
            "kernel_ms": kern,
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is reported at N=1 only
            state = {k: t_.detach().cpu().numpy() for k, t_ in net.state_dict().items()}
            res["cpu_baseline"] = cpu_baseline(c, state, SEED, args.cpu_budget)
    if dist.is_initialized():                                    # tear RCCL down first: the JSON line must be the last line of stdout
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
