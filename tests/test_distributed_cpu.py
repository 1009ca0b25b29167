"""world_size-2 gloo test of the N > 1 path of bench.py (CPU): the batch-sharded forward has no data-path collective, so
what must be right is (a) every rank draws a DIFFERENT shard of the global batch from the same parameter seed, (b) the
timing rule: barrier + sync on both sides, MAX over ranks, whole-job samples/s."""
import os
import sys
import time

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    c = dict(bench.C2, V=12, Q=3, A=5, v_dim=16, q_dim=8, a_dim=8)
    torch.manual_seed(bench.SEED)
    w = torch.randn(4)                                   # "parameters": identical on every rank
    v, _, _ = bench.synth_inputs(c, 6, bench.SEED + 1 + rank, "cpu")      # this rank's shard
    slow = 0.05 if rank == 1 else 0.01

    def step():
        time.sleep(slow)

    el = bench.measure(step, steps=4, warmup=1, world=world, sync=lambda: None, dist=dist, device="cpu")
    q.put((rank, el, w.tolist(), float(v.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, el0, w0, s0), (r1, el1, w1, s1) = res
    assert el0 == el1                                     # MAX over ranks: both ranks report the slow rank's time
    assert el0 >= 4 * 0.05 and el0 < 4 * 0.05 + 1.0
    assert w0 == w1                                       # same seed -> same parameters
    assert s0 != s1                                       # different shards of the global batch
    sys.path.insert(0, ROOT)
    import bench
    assert bench.whole_job_rate(2, 256, 4, el0) == 2 * 256 * 4 / el0


def test_flop_model_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    f = bench.flops_per_sample(bench.C2)
    assert f["tucker"] == 1051406336 and f["rank"] == 1666711552 and f["core_final"] == 3229728768
    assert abs(f["total"] / 1e6 - 5983.2) < 0.1           # 5 983.2 MFLOP per sample


def test_bare_multi_gpu_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (a torch.distributed.run child spawned before any
    GPU call), not exit: --dry-launch runs exactly that launch + rendezvous + barrier / max-reduce path on CPU (gloo)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "3", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    res = json.loads(last)                                   # the JSON line is the LAST line of stdout
    assert res == dict(res, dry_launch=True, n_gpus=2, ranks_seen=2, steps=3, warmup=1)
    assert res["ms_per_step"] >= 10.0


@pytest.mark.parametrize("extra", [[], ["--mode", "train"]])
def test_eight_rank_launch_and_device_binding_dry(extra):
    """`bench.py --gpus 8` (forward and --mode train) through its own launcher on CPU: eight ranks rendezvous over gloo at 127.0.0.1, time the
    barrier-bracketed steps, and rank r binds cuda:r (LOCAL_RANK -> device, what main() does before any kernel) -- BASELINE configs[4]'s launch shape
    (reference trainer: src/FFOE/trainer.py:221-232 has no launcher at all)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch", "--steps", "3", "--warmup", "1"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res == dict(res, dry_launch=True, n_gpus=8, ranks_seen=8, steps=3, warmup=1, mode="train" if extra else "forward")
    assert res["device_of_rank"] == ["cuda:%d" % i for i in range(8)]


def test_eight_ranks_under_the_drivers_own_launcher_dry():
    """... and launched the way the driver does it: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py
    --gpus 8 ...` -- the ranks read RANK / LOCAL_RANK / WORLD_SIZE from the environment and must NOT spawn ranks of their own."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29577",
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # ONE JSON line, from rank 0
    res = json.loads(lines[-1])
    assert res["ranks_seen"] == 8 and res["n_gpus"] == 8 and res["device_of_rank"] == ["cuda:%d" % i for i in range(8)]
