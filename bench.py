#!/usr/bin/env python3
"""bench.py -- CTI fused-forward samples/sec (BASELINE.json metric) on N MI355X GPUs of one node.

One "step" = one TCNet.forward (reference src/tc.py:41-52) over one batch of synthetic tensors of BASELINE.json
configs[1]: B=256 per GPU, V=36x2048, Q=14x1024, A=3129x300, rank=32, h_mm=512, glimpse=2, fp32 in / fp32 out,
inputs resident in HBM before the timed region.  The batch axis shards across GPUs with no data-path collective
(forward-only replicas, weak scaling: 256 rows per GPU); the only collectives are the timing barrier and the
max-over-ranks of the elapsed time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp32|bf16x3|f16f6|bf16] [--batch B] [--no-cpu-baseline]
    python bench.py --config c3|c4            full-model forwards of BASELINE configs[2] / [3] (bf16), their own flop tally and roofline
    python bench.py --mode train              data-parallel training step (configs[4] shape) of the CTI fusion block

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a `torch.distributed.run` child, spawned before
anything touches a GPU) and relays rank 0's JSON line; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it
is a rank.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the definition of every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs[1]
C2 = dict(B=256, V=36, Q=14, A=3129, v_dim=2048, q_dim=1024, a_dim=300, h_mm=512, rank=32, glimpse=2)
# MI355X_MICROARCH.md: f32 MFMA / dense bf16(f16) MFMA.  The fp32-grade split modes are priced against the 16-bit dense peak.
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16": 2500.0, "f16f6": 2500.0}
MFMA_UNITS_PER_PRODUCT = {"fp32": 1.0, "bf16x3": 3.0, "bf16": 1.0, "f16f6": 1.5}   # 16-bit-MFMA-equivalents issued per algorithmic product
HBM_PEAK_GBS = 8000.0
SEED = 1204                                                         # the reference's default seed (src/FFOE/main.py:53)
DTYPE_NAME = {"fp32": "f32", "bf16x3": "f32 (bf16x3 split products, f32 accumulate)", "bf16": "bf16 products, f32 accumulate",
              "f16f6": "f32 (f16 product + two block-scaled fp6 correction products, f32 accumulate)"}


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


def cpu_baseline(c, state, inputs, gpu_out=None, budget_s=20.0):
    """The oracle (numpy restatement of the reference's CPU path, oracle/cti_oracle.py) timed on this host's cores on a bounded sample of
    the SAME workload: the first B_cpu samples of rank 0's batch, repeated until ~budget_s of CPU work.  Because they are the samples the
    GPU just processed, the oracle's output also checks the timed launch (`gpu_out` = the HIP result for those samples): the normalised max
    error goes into the record and a miss of the north-star tolerance aborts the run instead of printing a number."""
    from oracle import cti_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    Bc = 4
    v, q, a = (t[:Bc].cpu().numpy() for t in inputs)
    O.tcnet_forward(v[:1], q[:1], a[:1], state)                  # warm-up (BLAS threads, page faults)
    t0 = time.perf_counter()
    n = 0
    while True:
        ref = O.tcnet_forward(v, q, a, state)
        n += Bc
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 1024:
            break
    rec = {"value": n / el, "unit": "samples/s", "cores": int(cores), "kind": "port",
           "sample": "oracle.tcnet_forward (numpy fp32; BLAS threads = cores, the einsum steps are single-threaded), %d samples of the C2 "
                     "shapes in batches of %d, %.1f s" % (n, Bc, el)}
    if gpu_out is not None:
        err = float(O.norm_max_err(gpu_out, ref))
        rec["parity_of_timed_launch"] = {"samples": Bc, "norm_max_err_vs_oracle": err, "tol": 1e-4}
        if not err < 1e-4:
            raise SystemExit("bench.py: the timed launch is %.3g from the oracle on its first %d samples (tolerance 1e-4) -- no number printed" % (err, Bc))
    return rec


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


def dist_info(dist, world, rank, dev):
    """What a multi-GPU line needs to be self-checking: how many ranks actually met in the RCCL communicator (an all-reduce of
    ones on the device), which physical GPU every rank ran on, and the RCCL version.  None without a process group."""
    if dist is None or not dist.is_initialized():
        return None
    ones = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(ones)
    pr = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "local_device": dev.index, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
            "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))}
    every = [None] * world
    dist.all_gather_object(every, mine)
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:                                                # pragma: no cover
        ver = "unknown (%s)" % type(e).__name__
    ids = {(d["uuid"], d["pci"]) for d in every}
    return {"ranks_seen": int(round(float(ones.item()))), "world_size": world, "distinct_devices": len(ids), "devices": every,
            "rccl_version": ver, "backend": dist.get_backend()}


def whole_job_rate(world, batch_per_rank, steps, elapsed):
    """samples/s of the whole job: every rank processed batch_per_rank * steps samples in `elapsed` (max over ranks)."""
    return world * batch_per_rank * steps / elapsed


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a torch.distributed.run CHILD (this process has not touched a GPU
    and never will), pass everything through, and finish with rank 0's JSON line as the last line of stdout and the child's return code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    last_json = None
    for line in p.stdout:
        if line.startswith("{") and line.rstrip().endswith("}"):
            last_json = line.rstrip()
        else:
            sys.stdout.write(line)
    rc = p.wait()
    sys.stdout.flush()
    if last_json is not None:
        print(last_json, flush=True)
    return rc


def run_dry(args, world, rank):
    """--dry-launch: the launch + rendezvous + timing path on CPU (gloo), no GPU and no kernels: every rank joins, sleeps through its
    "steps", and rank 0 reports how many ranks met at the barrier and which device each of them WOULD have bound (main() binds cuda:LOCAL_RANK) --
    so the first real 8-GPU run is not the first time `--gpus 8` executes."""
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    el = measure(lambda: time.sleep(0.01), args.steps, args.warmup, world, lambda: None, dist if world > 1 else None, "cpu")
    seen = torch.ones(1)
    binding = [(rank, "cuda:%d" % local)]
    if world > 1:
        dist.all_reduce(seen)
        got = [None] * world
        dist.all_gather_object(got, binding[0])
        binding = sorted(got)
        dist.barrier(); dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "mode": args.mode, "n_gpus": world, "ranks_seen": int(seen.item()), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": el / args.steps * 1e3, "device_of_rank": [d for _, d in binding]}), flush=True)


class CTIFusionBlock(torch.nn.Module):
    """The CTI fusion of CTIModel.forward (reference src/FFOE/base_model.py:128-135) at the VQA-2.0 shapes of BASELINE configs[3]/[4]:
    TriAttention (h_mm 512, rank 32, glimpse 2), one TCNet(k=2) pooling + q_prj/a_prj residual per glimpse, sum, and the reference's
    SimpleClassifier (src/classifier.py:19-28) -- all of it on the HIP library.  Used by --mode train only."""

    def __init__(self, cti, v_dim=2048, num_hid=1024, h_mm=512, rank=32, gamma=2, n_ans=3129):
        super().__init__()
        import types
        self.gamma = gamma
        self.t_att = cti.TriAttention(v_dim, num_hid, num_hid, h_mm, 1, rank, gamma, 1, dropout=[.2, .5])
        self.t_net = torch.nn.ModuleList([cti.TCNet(v_dim, num_hid, num_hid, h_mm, 1, rank, 1, dropout=[.2, .5], k=2) for _ in range(gamma)])
        self.q_prj = torch.nn.ModuleList([cti.FCNet([num_hid, num_hid], '', .2) for _ in range(gamma)])
        self.a_prj = torch.nn.ModuleList([cti.FCNet([num_hid, num_hid], '', .2) for _ in range(gamma)])
        self.classifier = cti.SimpleClassifier(num_hid, 2 * num_hid, n_ans, types.SimpleNamespace(activation="relu", dropout=0.5))

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
    forced = os.environ.get("CTI_BENCH_FORCE_DIST") == "1"
    opt = cti_amd.FlatAdamaxDP(model, lr=1e-3, clip_norm=0.25, force_collective=forced)
    opt.broadcast_parameters()
    crit = cti_amd.BCEWithLogitsSum()
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
        loss = crit(model(v, q, a), y) / B
        loss.backward()
        opt.step()

    # The step is captured (cti_amd.GraphedTrainStep) and replayed -- eagerly its ~300 launches take the host about as long to issue as the
    # GPU takes to run them, and the line would time the box's CPU.  One rank, no collective: ONE hipGraph.  With a process group (several
    # ranks, or CTI_BENCH_FORCE_DIST=1): TWO graphs -- forward + backward + gradient gather | clip + Adamax -- with the RCCL all-reduce issued
    # eagerly between them (capturing the collective itself aborted the process on this stack; --graph asks for that one-graph form anyway).
    graphed = not args.no_graph
    if graphed:
        gs = cti_amd.GraphedTrainStep(model, opt, lambda out, tgt: crit(out, tgt) / B, (v, q, a), y, warmup=3,
                                      split_collective=False if args.graph else None)
        step_fn = gs.replay                                      # (the inputs are resident: nothing to copy per step)
        launch = ("two hipGraphs (fwd + bwd + gather | clip + Adamax) around one eager RCCL all-reduce" if gs.split
                  else "hipGraph replay of the captured step")
    else:
        step_fn = step
        launch = "eager (one launch per kernel)"
    el = measure(step_fn, args.steps, args.warmup, world, torch.cuda.synchronize, dist, dev)
    # host time to ISSUE a step (no device sync inside the loop): what the ranks' CPUs spend per step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step_fn()
    host_ms = (time.perf_counter() - t0) / 10 * 1e3
    torch.cuda.synchronize()
    dinfo = dist_info(dist, world, rank, dev)
    ar = None
    if dist.is_initialized():
        # the step's ONE collective alone: the flat gradient buffer through RCCL, HIP events around the eager calls on the current stream;
        # bus bandwidth = bytes x 2 (N - 1) / N / time (what a ring all-reduce moves per link)
        flat = torch.zeros(opt.n_params, device=dev, dtype=torch.float32)
        for _ in range(3):
            dist.all_reduce(flat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); dist.barrier()
        e0.record()
        for _ in range(10):
            dist.all_reduce(flat)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        t = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        nbytes = opt.n_params * 4
        ar = {"bytes": nbytes, "ms": ms, "algbw_GBps": nbytes / ms / 1e6, "busbw_GBps": nbytes * 2 * (world - 1) / max(world, 1) / ms / 1e6,
              "note": "max over ranks of the mean of 10 eager all-reduces of the flat fp32 gradient buffer; busbw = bytes * 2 (N - 1) / N / time"}
        del flat
    if dist.is_initialized():                                   # tear RCCL down first: the JSON line must be the last line of stdout
        dist.barrier(); dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        coll = "rccl all-reduce of the flat gradient buffer" if world > 1 else ("rccl all-reduce executed (forced, world 1)" if forced else "none (one rank)")
        print(json.dumps({"metric": "CTI fusion-block data-parallel training samples/sec (256 rows/GPU, VQA-2.0 shapes)",
                          "value": whole_job_rate(world, B, args.steps, el), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
                          "config": {"workload": "BASELINE configs[4] shape: TriAttention + 2 x (TCNet.forward_with_weights, q_prj, a_prj) + SimpleClassifier "
                                                 "+ BCE, train mode (dropout on), fwd + bwd + one all-reduce + fused clip/Adamax",
                                     "global_batch": world * B, "parameters": opt.n_params, "parallelism": "dp%d" % world, "collective": coll,
                                     "launch": launch, "host_issue_ms_per_step": round(host_ms, 3)},
                          "distributed": dinfo, "allreduce": ar}))


# ---- full-model forwards: BASELINE configs[2] (c3: MC CTI, Visual7W shapes) and configs[3] (c4: FFOE BAN + CTI teacher) ---------------
def model_flops(kind, B, V, Q, A, G, vd=2048, nh=1024, h=512, R=32, n_ans=3129, rep=1, executed=False):
    """SURVEY.md 8(d) formulas summed over the module calls of the model forward (per BATCH, flops = 2 MAC).  `rep`: the MC pipeline feeds
    every image `rep` times; the reference projects v for every row, so the algorithmic count does too (the de-duplicated kernels do less)."""
    hr = h // R
    gru = lambda L, i: 2 * L * 3 * nh * (i + nh)                                   # noqa: E731  (input + recurrent products of a 1-layer GRU)
    cls = lambda o: 2 * (nh * 2 * nh + 2 * nh * o)                                 # noqa: E731
    if kind == "cti":
        tucker = 2 * h * (V * vd + Q * nh + A * nh)
        rank = 2 * h * h * (V + Q + A)
        core = 2 * V * R * hr * hr * G * hr + 2 * V * Q * R * hr * G * hr + 2 * V * Q * A * G * h
        d = 2 * h
        fww = G * (2 * d * (V * vd + Q * nh + A * nh) + 2 * d * (V * Q * A + V * Q + V))
        prj = G * 2 * 2 * nh * nh
        per = tucker + rank + core + fww + prj + gru(Q, 600) + gru(A, 600) + cls(n_ans)
        if executed:
            # what the de-duplicating kernels run: the terms that depend on the image alone once per image instead of once per row
            v_only = 2 * h * V * vd + 2 * h * h * V + 2 * V * R * hr * hr * G * hr + G * 2 * d * V * vd
            return B * (per - v_only) + (B // rep) * v_only
        return B * per
    if kind == "ban":
        d3 = 3 * nh
        att = 2 * d3 * (V * vd + Q * nh) + 2 * G * V * Q * d3 + G * V * d3
        bnet = G * (2 * nh * (V * vd + Q * nh) + 2 * nh * (V * Q + V))
        prj = G * 2 * nh * nh
        per = att + bnet + prj + gru(Q, 600) + cls(n_ans)
        return B * per
    raise ValueError(kind)


MODEL_TOL = {"bf16": 2e-2, "bf16x3": 1e-4, "f16f6": 1e-4, "fp32": 1e-4}     # normalised max error of the model logits vs the fp32 oracle, per arithmetic mode


IO_BF16 = "bf16 image features in; the projected v (hoisted projection GEMMs -> sum-pools / attention logits) as bf16 rows; token ids int64; q / a sequences and logits fp32"


def model_setup(config, B, rank, dev):
    """Model, synthetic batch and oracle hook of BASELINE configs[2] (c3: MC CTI, Visual7W shapes) / configs[3] (c4: FFOE BAN + CTI teacher).
    Returns dict(fwd, flops, workload, out_shape, oracle): fwd() -> logits tensor(s); oracle(n) -> list of (name, gpu rows, oracle rows) over the
    first n rows (oracle/cti_models.py, numpy fp32 -- a checker, after the timed region)."""
    import types
    import cti_amd
    ntoken = 20000
    torch.manual_seed(SEED)

    def ds(num_ans):
        return types.SimpleNamespace(dictionary=types.SimpleNamespace(ntoken=ntoken), v_dim=2048, num_ans_candidates=num_ans)

    def margs(gamma):
        return types.SimpleNamespace(op="c", num_hid=1024, gamma=gamma, h_mm=512, rank=32, k=1, h_out=1, activation="relu", dropout=0.5, use_counter=False)

    g = torch.Generator().manual_seed(SEED + 1 + rank)
    # configs[2] / [3] name bf16: in the plain-bf16 mode the image features -- 97 % of a batch's input bytes -- are handed over as bf16 (round 5); the oracle and
    # the other arithmetic modes see the same bf16-rounded values widened to fp32
    v_bf16 = cti_amd.get_precision() == "bf16" and os.environ.get("CTI_BENCH_V_FP32", "0") != "1"
    as_given = (lambda t: t.to(torch.bfloat16)) if v_bf16 else (lambda t: t)

    def tokens(L):
        t = torch.randint(0, ntoken, (B, L), generator=g)
        n = torch.randint(3, L + 1, (B,), generator=g)
        t[torch.arange(L)[None, :] >= n[:, None]] = ntoken
        return t.to(dev)

    def state(m):
        return {k: t_.detach().cpu().numpy() for k, t_ in m.state_dict().items()}

    if config == "c3":
        # Visual7W: 64 images x 4 candidate answers = 256 rows, every image repeated for its candidates (src/MC/train.py:75-79)
        rep = 4
        vi = torch.randn(B // rep, 36, 2048, generator=g).abs()
        nv = torch.randint(10, 37, (B // rep,), generator=g)
        vi[torch.arange(36)[None, :] >= nv[:, None]] = 0
        v = as_given(vi.unsqueeze(1).expand(-1, rep, -1, -1).contiguous().view(B, 36, 2048).to(dev))
        q1 = tokens(12)
        q = q1.view(B // rep, rep, 12)[:, :1].expand(-1, rep, -1).contiguous().view(B, 12)
        a = tokens(6)
        boxes = torch.rand(B, 36, 6, generator=g).to(dev)
        m = cti_amd.build_mc_cti(margs(2), ds(2)).to(dev).eval()
        # (TanModel.v_replication stays at its default 'auto': the repeated images are detected on the first forward and verified on the device on every later one)
        fwd = lambda: m(v, boxes, q, a)[0]                                           # noqa: E731

        def oracle(n, out, rows=None):
            from oracle import cti_models as OM
            ix = torch.arange(n) if rows is None else torch.as_tensor(rows)                # (rows: any subset of the batch -- the forward has no cross-row term)
            ref = OM.mc_tan_forward(v.cpu()[ix].float().numpy(), q.cpu()[ix].numpy(), a.cpu()[ix].numpy(), state(m), 2)[0]
            return [("mc_cti logits", out.cpu()[ix].numpy(), ref)]
        return dict(fwd=fwd, oracle=oracle, models={"mc_cti": m}, flops=model_flops("cti", B, 36, 12, 6, 2, n_ans=2), out_shape=(B, 2), rep=rep, v_bf16=v_bf16, inputs=dict(v=v, boxes=boxes, q=q, a=a),
                    executed_flops=model_flops("cti", B, 36, 12, 6, 2, n_ans=2, rep=rep, executed=True),
                    workload="BASELINE configs[2]: MC CTI model forward (TanModel, src/MC/base_model.py:128-152), Visual7W shapes, B=64 images x 4 candidates = %d rows, V=36, Q=12, A=6, glimpse 2" % B)
    v = torch.randn(B, 36, 2048, generator=g).abs()
    nv = torch.randint(10, 37, (B,), generator=g)
    v[torch.arange(36)[None, :] >= nv[:, None]] = 0
    v = as_given(v.to(dev))
    q, a = tokens(14), tokens(3)
    boxes = torch.rand(B, 36, 6, generator=g).to(dev)
    ban = cti_amd.build_ban(margs(8), ds(3129)).to(dev).eval()
    cti = cti_amd.build_cti(margs(2), ds(3129)).to(dev).eval()

    def fwd():
        # two models, one batch, nothing shared but the inputs: on sibling streams (ops.run_concurrently) unless CTI_BENCH_SERIAL_MODELS=1
        if os.environ.get("CTI_BENCH_SERIAL_MODELS", "0") == "1":
            return ban(v, boxes, q, None)[0], cti(v, q, a)
        if os.environ.get("CTI_BENCH_C4_ORDER", "cti_first") == "ban_first":               # (A/B knob: which model keeps the caller's stream and with it the auxiliary one)
            return cti_amd.ops.run_concurrently(lambda: ban(v, boxes, q, None)[0], lambda: cti(v, q, a))
        c, b = cti_amd.ops.run_concurrently(lambda: cti(v, q, a), lambda: ban(v, boxes, q, None)[0])
        return b, c

    def oracle(n, out, rows=None):
        from oracle import cti_models as OM
        ix = torch.arange(n) if rows is None else torch.as_tensor(rows)
        vn, qn, an = v.cpu()[ix].float().numpy(), q.cpu()[ix].numpy(), a.cpu()[ix].numpy()
        return [("ban logits", out[0].cpu()[ix].numpy(), OM.ffoe_ban_forward(vn, qn, state(ban), 8)[0]),
                ("cti logits", out[1].cpu()[ix].numpy(), OM.ffoe_cti_forward(vn, qn, an, state(cti), 2))]
    return dict(fwd=fwd, oracle=oracle, models={"ban": ban, "cti": cti}, flops=model_flops("ban", B, 36, 14, 0, 8) + model_flops("cti", B, 36, 14, 3, 2), out_shape=(B, 3129), v_bf16=v_bf16, inputs=dict(v=v, boxes=boxes, q=q, a=a),
                workload=("BASELINE configs[3]: FFOE teacher forward = BanModel (BiAttention glimpse 8, src/FFOE/base_model.py:37-67) + CTIModel (glimpse 2, "
                          ":112-136), VQA-2.0 shapes, B=%d, V=36, Q=14, A=3, 3129 classes; %s" % (
                              B, "one model after the other" if os.environ.get("CTI_BENCH_SERIAL_MODELS", "0") == "1" else
                              "the two models on sibling streams (ops.run_concurrently; CTI_BENCH_SERIAL_MODELS=1 = one after the other)")))


def model_measure(setup, steps, warmup, world, dist, dev, graphed, prec):
    """Times setup['fwd'] (hipGraph replay unless graphed is False) and checks the forward that was TIMED (the replayed graph's static output)
    against the oracle on its first 4 rows.  Returns (elapsed, parity record)."""
    from oracle.cti_oracle import norm_max_err
    holder = {}

    def step():
        holder["out"] = setup["fwd"]()

    # The forward is ~190 launches for 1.3-3.2 ms of GPU work: issued eagerly it times the box's CPU as much as the GPU (74 k and 81 k samples/s
    # for c4 on two boxes).  It contains no collective, allocates nothing and never synchronises, so it is captured into a hipGraph once and
    # replayed (tools/graph_model.py checks replay == eager); --no-graph keeps the eager launches.
    with torch.no_grad():
        if graphed:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(gr, stream=side):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            el = measure(gr.replay, steps, warmup, world, torch.cuda.synchronize, dist, dev)
        else:
            el = measure(step, steps, warmup, world, torch.cuda.synchronize, dist, dev)
    out = holder["out"]
    out0 = out[0] if isinstance(out, tuple) else out
    assert tuple(out0.shape) == setup["out_shape"] and bool(torch.isfinite(out0).all())
    parity = {"rows": 4, "tol": MODEL_TOL[prec], "vs": "oracle/cti_models.py (numpy fp32) on the first 4 rows of the timed forward's output"}
    for name, got, ref in setup["oracle"](4, out):
        parity[name] = float(norm_max_err(got, ref))
        if not parity[name] < MODEL_TOL[prec]:
            raise SystemExit("bench.py: %s of the timed %s forward are %.3g from the oracle (tolerance %.3g) -- no number printed" % (name, prec, parity[name], MODEL_TOL[prec]))
    # ... and EVERY row of the timed forward against the library's own fp32-grade (bf16x3) forward of the same model and batch, on the device: the oracle
    # leg covers four samples, and a geometry bug that zeroed the last rows of every 256-row GEMM tile once sat outside them
    if prec != "bf16x3":
        import cti_amd
        timed = [t.clone() for t in (out if isinstance(out, (tuple, list)) else (out,))]
        cti_amd.set_precision("bf16x3")
        try:
            with torch.no_grad():
                ref_out = setup["fwd"]()
        finally:
            cti_amd.set_precision(prec)
        ref_out = ref_out if isinstance(ref_out, (tuple, list)) else (ref_out,)
        worst = max(float(((a.float() - b.float()).abs().flatten(1).amax(1) / b.float().abs().max()).max()) for a, b in zip(timed, ref_out))
        parity["every_row_vs_bf16x3_forward"] = worst
        if not worst < MODEL_TOL[prec]:
            raise SystemExit("bench.py: a row of the timed %s forward is %.3g from the bf16x3 forward (tolerance %.3g) -- no number printed" % (prec, worst, MODEL_TOL[prec]))
    return el, parity


def run_model(args, world, rank, dev, dist):
    """--config c3 | c4: eval forward of the full models at BASELINE configs[2] / [3] in plain-bf16 products (the dtype those configs name)."""
    import cti_amd
    prec = args.precision if args.precision_given else "bf16"
    cti_amd.set_precision(prec)
    B = args.batch
    setup = model_setup(args.config, B, rank, dev)
    graphed = not args.no_graph
    el, parity = model_measure(setup, args.steps, args.warmup, world, dist, dev, graphed, prec)
    flops, workload = setup["flops"], setup["workload"]
    dinfo = dist_info(dist, world, rank, dev)
    if dist.is_initialized():
        dist.barrier(); dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        ach = flops * args.steps / el / 1e12
        peak = PEAK_TFLOPS[prec]
        print(json.dumps({
            "metric": "full-model forward samples/sec (%s)" % args.config, "value": whole_job_rate(world, B, args.steps, el), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[prec], "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * B, "precision": prec, "parallelism": "replicas x%d" % world,
                       "gflop_per_batch": round(flops / 1e9, 3), "executed_gflop_per_batch": round(setup.get("executed_flops", flops) / 1e9, 3),
                       "io_dtype": ("%s; products in %s" % (IO_BF16, prec)) if setup.get("v_bf16") else "fp32 activations in and out (token ids int64); products in %s" % prec,
                       "launch": "hipGraph replay of the captured forward" if graphed else "eager"},
            "roofline": {"bound": "mfma", "kernel": "whole forward (launch sequence; dominant kernels are the projection GEMMs)", "achieved": ach,
                         "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                         "note": "algorithmic flops of SURVEY.md 8(d) summed over the module calls / wall time of the forward"},
            "parity_of_timed_forward": parity, "distributed": dinfo}), flush=True)


def model_subrecord(config, dev):
    """The c3 / c4 full-model forward as a sub-record of the default line: bf16 (the dtype configs[2] / [3] name), hipGraph replay, ~0.3 s."""
    import cti_amd
    old = cti_amd.get_precision()
    cti_amd.set_precision("bf16")
    try:
        setup = model_setup(config, 256, 0, dev)
        steps = 200 if config == "c3" else 100
        el, parity = model_measure(setup, steps, 10, 1, None, dev, True, "bf16")
        as_called = None
        if setup.get("rep", 1) > 1:
            # the timed forward is what a drop-in caller gets (src/MC/train.py:75-79 replicates the image rows and says nothing): v_replication = 'auto'.
            # Beside it: the explicit hint (no per-forward check of the batch) and de-duplication off (every row's image projected)
            as_called = {}
            for name, val, note in (("hint", setup["rep"], "TanModel.v_replication = %d set by the caller: no detection, no per-batch check" % setup["rep"]),
                                    ("off", 1, "TanModel.v_replication = 1: every row's image projected")):
                for m in setup["models"].values():
                    m.v_replication = val
                el1, _ = model_measure(setup, steps, 10, 1, None, dev, True, "bf16")
                as_called[name] = {"v_replication": val, "value": 256 * steps / el1, "unit": "samples/s", "ms_per_step": el1 / steps * 1e3,
                                   "achieved_tflops": setup["flops"] * steps / el1 / 1e12, "note": note}
            for m in setup["models"].values():
                m.v_replication = "auto"
    finally:
        cti_amd.set_precision(old)
    ach = setup["flops"] * steps / el / 1e12
    exe = setup.get("executed_flops", setup["flops"]) * steps / el / 1e12
    rec = {"workload": setup["workload"], "value": 256 * steps / el, "unit": "samples/s", "ms_per_step": el / steps * 1e3, "steps": steps,
           "dtype": DTYPE_NAME["bf16"], "io_dtype": (IO_BF16 + "; bf16 products") if setup.get("v_bf16") else "fp32 activations in and out; bf16 products",
           "launch": "hipGraph replay of the captured forward",
           "gflop_per_batch": round(setup["flops"] / 1e9, 3), "executed_gflop_per_batch": round(setup.get("executed_flops", setup["flops"]) / 1e9, 3),
           "achieved_tflops": ach, "frac_of_bf16_peak": ach / PEAK_TFLOPS["bf16"],
           "executed_tflops": exe, "executed_frac_of_bf16_peak": exe / PEAK_TFLOPS["bf16"], "parity_of_timed_forward": parity}
    if as_called is not None:
        rec["v_replication"] = "auto (detected %d: the timed forward is the reference's own calling convention)" % setup["rep"]
        rec["other_v_replication_settings"] = as_called
    return rec


def projection_gemm_record(dev):
    """The projection GEMMs of the bf16 configurations alone (north star: >= 0.40 of the bf16 MFMA peak on the projection GEMMs), through cti_gemm_bf16_rows
    (csrc/cti_gemm16.hip): bf16 rows in, bias + ReLU, bf16 / fp32 rows out; random operands; every row checked against float64 on a strided sample of columns.
    Two shapes: the hoisted v projections of BASELINE configs[2] (CTI: 9 216 rows = 256 x 36 objects, K = 2 048, three 1 024-wide layers -- 432 tiles = 1.69 rounds
    of 256 workgroups) and of configs[3] (BAN: eight 1 024-wide layers -- 1 152 tiles = 4.5 rounds)."""
    import cti_amd
    ops = cti_amd.ops
    rec = {"kernel": "gemm16_planes_kernel (cti_gemm_bf16_rows / _sk): plain-bf16 NT GEMM, 256 x 256 tile, two wave groups one interval apart; stream-K cut from three rounds of tiles (the second shape)",
           "bound": "mfma", "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s", "shapes": {}}
    for M, N, K in ((9216, 3072, 2048), (9216, 8192, 2048)):
        g = torch.Generator(device="cpu").manual_seed(SEED + 11)
        a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g) / 8).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        wp = ops.split_operand(w, prec="bf16")
        out = {}
        for name, dt in (("bf16_out", torch.bfloat16), ("fp32_out", torch.float32)):
            fn = lambda: ops.gemm_bf16_rows(a, wp, N, out_dtype=dt, bias=b, relu=True)      # noqa: E731
            for _ in range(60):                                           # ~7-20 ms of back-to-back launches first: the first rounds after an idle spell read 5-25 % slow
                y = fn()                                                  # (119.6 / 141.0 / 127.9 us, then 113-116 for good: gpurun_out r06, same box, same process)
            rounds = []
            for _ in range(5):                                            # five rounds of 20 back-to-back launches: the median round counts (min beside it)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(20):
                    y = fn()
                e1.record(); torch.cuda.synchronize()
                rounds.append(e0.elapsed_time(e1) / 20 * 1e3)
            us = sorted(rounds)[2]
            cols = torch.arange(0, N, 97, device=dev)
            ref = torch.relu(a.double() @ w.to(torch.bfloat16).double()[cols].t() + b.double()[cols])
            err = float((y[:, cols].double() - ref).abs().max() / ref.abs().max())
            tol = 4e-3 if dt == torch.bfloat16 else 3e-6
            if not err < tol:
                raise SystemExit("bench.py: projection GEMM %dx%dx%d (%s) is %.3g from float64 (tolerance %.0e) -- no number printed" % (M, N, K, name, err, tol))
            tf = 2.0 * M * N * K / us * 1e-6
            out[name] = {"us": us, "tflops": tf, "frac_of_bf16_peak": tf / PEAK_TFLOPS["bf16"], "us_best_round": min(rounds),
                         "norm_max_err_every_row_sampled_columns_vs_float64": err}
        rec["shapes"]["%dx%dx%d" % (M, N, K)] = out
        del a, w, b, wp, y
    first = rec["shapes"]["9216x3072x2048"]
    rec["shape"] = "9216 x 3072 x 2048 (bf16 rows x resident weight planes, bias + ReLU epilogue)"       # (the round-3 / early round-4 layout of this record: the first shape)
    rec["bf16_out"], rec["fp32_out"] = first["bf16_out"], first["fp32_out"]
    return rec


def aside_kernels(c, dev):
    """The a-side kernels of the f16f6 step, launched stand-alone at the configs[1] shapes through their own C-ABI entry points and timed with
    HIP events on the launch stream: inside cti_tcnet_forward they run back to back on the main stream (only the mode-3 product has events of
    its own).  -> list of roofline_kernels records."""
    import cti_amd
    ops, L = cti_amd.ops, cti_amd.pkg._lib
    lib = L.lib()
    rows, K1, h, A = c["B"] * c["A"], c["a_dim"], c["h_mm"], c["A"]
    st = ops._stream()

    def t(fn, n=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    g = torch.Generator(device="cpu").manual_seed(SEED + 7)
    x = torch.randn(rows, K1, generator=g).to(dev)
    w1 = (torch.randn(h, K1, generator=g) * 0.05).to(dev)
    w2 = (torch.randn(h, h, generator=g) * 0.05).to(dev)
    b1 = torch.randn(h, generator=g).to(dev) * 0.1
    nbx = lib.cti_f16f6_planes_bytes(rows, K1, 0)
    px = torch.empty(nbx, device=dev, dtype=torch.uint8)
    L.check(lib.cti_quantize_f16f6(x.data_ptr(), K1, rows, K1, 0, px.data_ptr(), nbx, st), "cti_quantize_f16f6")      # (zero-fills the block's slack rows once)
    # the pass as cti_tcnet_forward launches it: quantize_rows_f16f6_kernel alone (round 4 timed the C-ABI call's 0.72 GB memset with it: 0.45 vs 0.32 ms in the step)
    t_q = t(lambda: L.check(lib.cti_quantize_f16f6_into(x.data_ptr(), K1, rows, K1, 0, px.data_ptr(), nbx, st), "cti_quantize_f16f6_into"))
    pw1, pw2 = ops.quantize_f16f6(w1), ops.quantize_f16f6(w2)
    nby = lib.cti_f16f6_planes_bytes(rows, h, 0)
    y1 = torch.zeros(nby, device=dev, dtype=torch.uint8)
    t_t = t(lambda: L.check(lib.cti_gemm_nt_f16f6_planes(pw1.data_ptr(), h, px.data_ptr(), rows, y1.data_ptr(), nby, 0, h, rows, K1, b1.data_ptr(), 1, st), "tucker"))
    nbz = lib.cti_f16f6_planes_bytes(rows, h, A)
    y2 = torch.zeros(nbz, device=dev, dtype=torch.uint8)
    t_r = t(lambda: L.check(lib.cti_gemm_nt_f16f6_planes(pw2.data_ptr(), h, y1.data_ptr(), rows, y2.data_ptr(), nbz, A, h, rows, h, b1.data_ptr(), 1, st), "rank"))
    q_bytes = rows * K1 * 4 + rows * ((K1 + 31) // 32) * 90
    peak = PEAK_TFLOPS["f16f6"]
    recs = [{"kernel": "quantize_rows_f16f6_kernel (`a` fp32 -> f16f6 planes; the step's own variant, no memset)", "ms": t_q, "bound": "hbm", "achieved": q_bytes / t_q / 1e6, "unit": "GB/s",
             "frac": q_bytes / t_q / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes": q_bytes}]
    for name, ms, K in (("gemm_f16f6_kernel<EPI_PLANES_T> a-side Tucker projection (512 x %d x %d)" % (rows, K1), t_t, K1),
                        ("gemm_f16f6_kernel<EPI_PLANES_T> a-side rank nets (512 x %d x 512)" % rows, t_r, h)):
        fl = 2.0 * rows * h * K
        recs.append({"kernel": name, "ms": ms, "bound": "mfma", "achieved": fl / ms / 1e9, "unit": "TFLOP/s", "frac": fl / ms / 1e9 / peak, "flops": fl})
    return recs


def run_forward(args, world, rank, dev, dist):
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
    assert out.shape == (c["B"], c["V"], c["Q"], c["A"], c["glimpse"])
    fp32_exact = None
    if rank == 0 and world == 1 and args.precision != "fp32" and not args.no_fp32_exact:
        # the strict-fp32 arithmetic (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain) on the driver's clock, a few steps
        gpu_first = out[:4].cpu().numpy()
        timed_out = out                                       # kept until the bf16x3 leg below has looked at every sample of it (3.2 GB of the 288)
        out = None
        del res_holder["out"]
        cti_amd.set_precision("fp32")
        with torch.no_grad():
            el32 = measure(step, 4, 1, 1, torch.cuda.synchronize, None, dev)
            o32 = res_holder["out"][:4].cpu().numpy()
            # every sample of the TIMED launch against the exact-fp32 launch (independent kernels end to end), on the device
            o32f = res_holder["out"]
            worst32, scale32 = 0.0, float(o32f.abs().max())
            for i0 in range(0, c["B"], 16):
                worst32 = max(worst32, float((timed_out[i0:i0 + 16] - o32f[i0:i0 + 16]).abs().max()))
            worst32 /= scale32
            tol32 = 2e-2 if args.precision == "bf16" else 1e-4
            if not worst32 < tol32:
                raise SystemExit("bench.py: a sample of the timed %s launch is %.3g from the exact-fp32 launch (tolerance %.0e) -- no number printed" % (args.precision, worst32, tol32))
            o32f = None
        bf16x3 = None
        if args.precision != "bf16x3":                      # the 3-term split-bf16 mode on the same clock, a few steps
            cti_amd.set_precision("bf16x3")
            with torch.no_grad():
                elx = measure(step, 10, 3, 1, torch.cuda.synchronize, None, dev)
                ox = res_holder["out"][:4].cpu().numpy()
            bf16x3 = {"value": c["B"] * 10 / elx, "unit": "samples/s", "ms_per_step": elx / 10 * 1e3, "steps": 10,
                      "norm_max_diff_vs_exact_fp32_first_4_samples": float(np.max(np.abs(ox - o32)) / np.max(np.abs(o32)))}
            # whole-launch coverage of the TIMED launch: every sample of the default mode's output against this mode's, on the device (the oracle leg
            # looks at four samples; the persistent tile walk of the other 252 is under this)
            oxf = res_holder["out"]
            worst, scale_ = 0.0, float(oxf.abs().max())
            for i0 in range(0, c["B"], 16):
                worst = max(worst, float((timed_out[i0:i0 + 16] - oxf[i0:i0 + 16]).abs().max()))
            bf16x3["every_sample_max_diff_of_the_timed_launch_vs_this_mode"] = worst / scale_
            tol_all = 2e-2 if args.precision == "bf16" else 1e-4
            if not worst / scale_ < tol_all:
                raise SystemExit("bench.py: a sample of the timed %s launch is %.3g from the bf16x3 launch (tolerance %.0e) -- no number printed" % (args.precision, worst / scale_, tol_all))
            oxf = None
            del res_holder["out"]
        timed_out = None
        cti_amd.set_precision(args.precision)
        fl = flops_per_sample(c)
        fp32_exact = {"value": c["B"] * 4 / el32, "unit": "samples/s", "ms_per_step": el32 / 4 * 1e3, "steps": 4,
                      "whole_step_tflops": fl["total"] * c["B"] * 4 / el32 / 1e12, "frac_of_f32_mfma_peak": fl["total"] * c["B"] * 4 / el32 / 1e12 / PEAK_TFLOPS["fp32"],
                      "norm_max_diff_of_default_mode_vs_exact_fp32_first_4_samples": float(np.max(np.abs(gpu_first - o32)) / np.max(np.abs(o32))),
                      "every_sample_max_diff_of_the_timed_launch_vs_this_mode": worst32}
        res_holder.pop("out", None)
    else:
        bf16x3 = None
        gpu_first = out[:4].cpu().numpy() if rank == 0 else None
    res = None
    if rank == 0:
        fl = flops_per_sample(c)
        core_ms = float(np.mean(kt.get("paralind_core", kt.get("tcnet_forward"))))
        core_flops = fl["core_final"] * c["B"]                   # algorithmic flops of ONE launch of the dominant kernel
        achieved = core_flops / (core_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        kern = {k: round(float(np.mean(ms)), 3) for k, ms in sorted(kt.items())}
        # HBM bytes of the dominant kernel come from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled
        # as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is); the summary is committed under profiles/ and names the kernel
        # variant it was taken on -- a record for another variant / batch is refused (traffic = null) rather than quoted stale.
        traffic, traffic_src, pmc_extra = None, None, {}
        tf = os.path.join(ROOT, "profiles", "core_traffic.json")
        variant = core_variant(args.precision)
        if os.path.isfile(tf):
            tj = json.load(open(tf))
            if tj.get("variant") == variant and tj.get("batch") == c["B"]:
                traffic, traffic_src = tj["hbm_bytes_per_launch"], tj["source"]
                pmc_extra = {k: tj[k] for k in ("l2_hit_rate", "mfma_busy_frac", "effective_clock_ghz", "read_bytes", "write_bytes", "lds_bank_conflict_frac") if k in tj}
            else:
                traffic_src = "profiles/core_traffic.json is for variant %r at B=%r, this run is %r at B=%d: not quoted" % (tj.get("variant"), tj.get("batch"), variant, c["B"])
        units = MFMA_UNITS_PER_PRODUCT[args.precision]
        res = {
            "metric": "CTI fused-forward samples/sec at B=256 (V=36x2048)",
            "value": whole_job_rate(world, c["B"], args.steps, el), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: TCNet.forward (fp32 in / fp32 out; arithmetic: see dtype), B=%d/GPU, V=36x2048, Q=14x1024, A=3129x300, rank=32, "
                                   "h_mm=512, glimpse=2" % c["B"], "global_batch": world * c["B"], "precision": args.precision,
                       "parallelism": "replicas x%d (batch-sharded, no data-path collective)" % world,
                       "gflop_per_sample": round(fl["total"] / 1e9, 4)},
            "roofline": {"bound": "mfma", "kernel": "paralind_core (mode-3 product + rank sum, batched NT GEMM 1008x3129x512 x%d): %s" % (c["B"], variant),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "launch_ms": core_ms, "flops_per_launch": core_flops,
                         "mfma_issued_tflops": achieved * units, "mfma_issued_frac": achieved * units / peak,
                         "algorithmic_bytes_per_launch": c["B"] * 4 * (c["V"] * c["Q"] * c["A"] * c["glimpse"] + c["h_mm"] * (c["V"] * c["Q"] * c["glimpse"] + c["A"])),
                         "traffic_source": traffic_src,
                         "note": "achieved = algorithmic fp32 flops / launch time; the split modes issue %.1f 16-bit-MFMA-equivalents per product, so the "
                                 "MFMA pipes run at mfma_issued_tflops against the 2500 TFLOP/s dense 16-bit peak" % units},
            "whole_step_tflops": fl["total"] * c["B"] * args.steps / el / 1e12,
            "kernel_ms": kern,
        }
        if args.precision == "f16f6":
            # counters of the same kernel variant from the committed PMC passes, and the bytes its tile walk moves from L2 into LDS by
            # construction: every 256 x 192 tile DMAs (256 + 192) rows x K x 2.8125 B (f16 + fp6 codes + scale bytes)
            tiles = c["B"] * (-(-c["V"] * c["Q"] * c["glimpse"] // 256)) * (-(-c["A"] // 192))
            res["roofline"].update(pmc_extra)
            res["roofline"]["l2_to_lds_bytes"] = int(tiles * (256 + 192) * c["h_mm"] * 2.8125)
            res["roofline"]["l2_to_lds_note"] = "derived: tiles x (256 + 192) operand rows x K x 2.8125 B per launch; l2_hit_rate / mfma_busy_frac: PMC passes named in traffic_source"
        rk = [{"kernel": "gemm_f16f6_kernel<EPI_INTERLEAVE2> mode-3 product + rank sum" if args.precision == "f16f6" else "mode-3 product + rank sum: " + variant,
               "ms": core_ms, "bound": "mfma", "achieved": achieved, "unit": "TFLOP/s", "frac": achieved / peak, "flops": core_flops}]
        if args.precision == "f16f6" and world == 1 and not args.no_subrecords:
            out = None
            res_holder.pop("out", None)
            torch.cuda.empty_cache()
            rk += aside_kernels(c, dev)
        step_ms = el / args.steps * 1e3
        for r in rk:
            r["share_of_step"] = r["ms"] / step_ms
        res["roofline_kernels"] = {"note": "every kernel family >= 5 % of the step: mode-3 from the library's hipEvents inside the timed steps; the "
                                           "a-side kernels re-launched stand-alone at the same shapes (HIP events on the launch stream)", "kernels": rk}
        if args.precision == "f16f6":
            # what the range / cancellation guard saw over this run's guarded launches (the host waits for every verdict in this eager loop: a tripped call
            # would have been re-run as bf16x3 / fp32 inside the timed region and counted here)
            gs = cti_amd.ops.f16f6_range_status()
            res["guard"] = {"guarded_calls": gs["calls"], "trips": gs["trips"], "last_status": gs["last_status"], "mode": "sync (host reads every verdict)"}
        if fp32_exact is not None:
            res["fp32_exact"] = fp32_exact
        if bf16x3 is not None:
            res["bf16x3"] = bf16x3
        if world == 1 and not args.no_subrecords and args.batch == C2["B"]:
            # BASELINE configs[2] / [3] on the same clock: full-model forwards, bf16, hipGraph replay, each checked against the oracle
            torch.cuda.empty_cache()
            res["configs"] = {"c3": model_subrecord("c3", dev), "c4": model_subrecord("c4", dev)}
            res["projection_gemm"] = projection_gemm_record(dev)
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is reported at N=1 only
            state = {k: t_.detach().cpu().numpy() for k, t_ in net.state_dict().items()}
            res["cpu_baseline"] = cpu_baseline(c, state, (v, q, a), gpu_first, args.cpu_budget)
    dinfo = dist_info(dist, world, rank, dev)
    if dist.is_initialized():                                    # tear RCCL down first: the JSON line must be the last line of stdout
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        res["distributed"] = dinfo
        print(json.dumps(res), flush=True)


def core_variant(precision):
    """Name of the mode-3 GEMM instantiation a precision mode selects at the configs[1] shape (what profiles/core_traffic.json must match)."""
    return {"bf16x3": "gemm_planes_kernel<terms=3, epi=INTERLEAVE2, tile 256x256, 4-slot ring>",
            "bf16": "gemm_planes_kernel<terms=1, epi=INTERLEAVE2, tile 256x256, 4-slot ring>",
            "fp32": "gemm_nt_f32_kernel<128x128x32>",
            "f16f6": "gemm_f16f6_kernel<epi=INTERLEAVE2, tile 256x192>"}[precision]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="forward", choices=["forward", "train"], help="forward (the BASELINE metric) or train (DP step)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4"], help="c2 = BASELINE configs[1] (the metric); c3 / c4 = full-model forwards of configs[2] / [3]")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (about 1 s of GPU time at the default workload: long enough for the clocks to settle; + ~20 s of CPU baseline)")
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps: allocator growth, one-time kernel attributes, clock ramp")
    ap.add_argument("--batch", type=int, default=C2["B"], help="rows per GPU (default 256 = BASELINE configs[1])")
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16x3", "bf16", "f16f6"],
                    help="f16f6 (default of the headline line): mode-3 product as f16 hi x hi + one block-scaled fp6 MFMA for both cross terms, every "
                         "other GEMM bf16x3 -- fp32-grade (3e-5 vs the float64 oracle at the configs[1] shape, tolerance 1e-4); bf16x3: 3-term split-bf16 "
                         "everywhere (1.5e-5); fp32: exact fp32 MFMA")
    ap.add_argument("--graph", action="store_true", help="--mode train with a process group: capture the whole step INCLUDING the all-reduce into one hipGraph (aborts on ROCm 7.0 / torch 2.10; the default there is two graphs around an eager all-reduce)")
    ap.add_argument("--no-graph", action="store_true", help="--mode train: eager launches even on one rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-exact", action="store_true", help="skip the 4-step exact-fp32 sub-record of the default line")
    ap.add_argument("--no-subrecords", action="store_true", help="skip the c3 / c4 model sub-records and the stand-alone a-side kernel timings of the default line")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--dry-launch", action="store_true", help="launch + rendezvous + timing path only (gloo on CPU, no GPU work)")
    args = ap.parse_args()
    args.precision_given = args.precision is not None
    if args.precision is None:
        args.precision = os.environ.get("CTI_PRECISION", "f16f6" if (args.mode == "forward" and args.config == "c2") else "bf16x3")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))         # before anything touches a GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if args.dry_launch:
        return run_dry(args, world, rank)
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
    elif args.config != "c2":
        run_model(args, world, rank, dev, dist)
    else:
        run_forward(args, world, rank, dev, dist)
    if dist.is_initialized():
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
