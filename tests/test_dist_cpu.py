"""N>1 path on CPU: world_size-2 gloo run of the prompt sharding / relevance all-gather /
weight broadcast (lxt_amd.dist).  The per-prompt work is replaced by a pure function of the
ids so the test pins exactly what the collectives must preserve: global prompt order and
rank-sharded == single-process results."""
import os
import subprocess
import sys
import textwrap

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    import lxt_amd.dist as D
    import torch.distributed as dist
    rank, world, _ = D.init(backend="gloo")
    assert world == 2
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 1000, (7, 16), generator=g)          # 7 prompts: uneven shards (4 + 3)
    fake = lambda x: (x.float() * 0.5 + x.float().cumsum(1))     # stands in for engine.explain
    R = D.explain_sharded(fake, ids, batch=2)
    assert R.shape == (7, 16) and torch.equal(R, fake(ids)), "sharded != single-process"
    w = [torch.full((5, 3), float(rank + 1)), torch.arange(4.0) * (rank + 1)]
    D.broadcast_weights(w, src=0)
    assert torch.equal(w[0], torch.full((5, 3), 1.0)) and torch.equal(w[1], torch.arange(4.0))
    lo, hi = D.shard_range(7, rank, world)
    assert (lo, hi) == ((0, 4) if rank == 0 else (4, 7))
    dist.barrier()
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ok_rank%%d" %% rank), "w").write("ok")   # (stdout of two ranks interleaves)
""") % ROOT


def test_sharding_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for attempt in range(3):                              # the rendezvous port can be taken between probe and bind: retry
        with socket.socket() as sk:                       # a free port: a fixed one collides with TIME_WAIT leftovers
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        if r.returncode == 0 or "sharded != single-process" in (r.stdout + r.stderr):
            break                                         # success, or a REAL failure of the thing under test
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok_rank0").exists() and (tmp_path / "ok_rank1").exists(), r.stdout + r.stderr


def test_shard_range_partitions():
    import lxt_amd.dist as D
    for n in (0, 1, 7, 8, 1024):
        for world in (1, 2, 4, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launch_contract_dry_run():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2 --steps K --warmup W` (the driver's
    launch line) through bench.py's control flow with the kernels stubbed out: exactly one JSON line, from rank 0, with the
    contract's keys; and the single-process form"""
    import json
    import socket
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for attempt in range(3):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        if r.returncode == 0 or "global prompt order" in (r.stdout + r.stderr):
            break
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config"):
        assert key in d
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["dry_run"] is True
    r1 = subprocess.run([sys.executable, bench, "--steps", "1", "--warmup", "0", "--dry-run"], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1
