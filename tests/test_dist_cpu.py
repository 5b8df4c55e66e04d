"""N>1 path on CPU: world_size-2 and -4 gloo runs of the prompt sharding / relevance all-gather /
weight broadcast (lxt_amd.dist).  The per-prompt work is replaced by a pure function of the
ids so the test pins exactly what the collectives must preserve: global prompt order and
rank-sharded == single-process results."""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, %r)
    import lxt_amd.dist as D
    import torch.distributed as dist
    rank, world, _ = D.init(backend="gloo")
    assert world == int(sys.argv[1])
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 1000, (7, 16), generator=g)          # 7 prompts: uneven shards (4 + 3 / 2 + 2 + 2 + 1)
    fake = lambda x: (x.float() * 0.5 + x.float().cumsum(1))     # stands in for engine.explain
    R = D.explain_sharded(fake, ids, batch=2)
    assert R.shape == (7, 16) and torch.equal(R, fake(ids)), "sharded != single-process"
    w = [torch.full((5, 3), float(rank + 1)), torch.arange(4.0) * (rank + 1)]
    D.broadcast_weights(w, src=0)
    assert torch.equal(w[0], torch.full((5, 3), 1.0)) and torch.equal(w[1], torch.arange(4.0))
    D.BCAST_CHUNK = 5                                             # the chunked form (flat buffers beyond 2^30 elements): 17 elements in pieces of 5
    big = [torch.arange(17.0).view(17) * (rank + 1), torch.full((2, 3), float(rank))]
    D.broadcast_weights(big, src=0)
    assert torch.equal(big[0], torch.arange(17.0)) and torch.equal(big[1], torch.zeros(2, 3))
    D.BCAST_CHUNK = 1 << 30
    lo, hi = D.shard_range(7, rank, world)
    assert (lo, hi) == ({2: [(0, 4), (4, 7)], 4: [(0, 2), (2, 4), (4, 6), (6, 7)]}[world][rank])
    # replica self-check: equal replicas pass; a replica whose words arrived in another ORDER (same multiset: the round-4 plain sum could not
    # tell) fails on every rank
    flat = torch.arange(4096, dtype=torch.float32).to(torch.bfloat16)
    assert len(set(D.check_replicas([flat, flat[:7]]))) == 1
    bad = flat.flip(0) if rank == world - 1 else flat
    try:
        D.check_replicas([bad])
        raise SystemExit("a permuted replica went unnoticed")
    except AssertionError:
        pass
    D.check_gather_order(7, 16, "cpu")
    dist.barrier()
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ok_rank%%d" %% rank), "w").write("ok")   # (stdout of two ranks interleaves)
""") % ROOT





@pytest.mark.parametrize("world", [2, 4])
def test_sharding_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for attempt in range(3):                              # the rendezvous port can be taken between probe and bind: retry
        with socket.socket() as sk:                       # a free port: a fixed one collides with TIME_WAIT leftovers
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(script), str(world)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        if r.returncode == 0 or "sharded != single-process" in (r.stdout + r.stderr):
            break                                         # success, or a REAL failure of the thing under test
    assert r.returncode == 0, r.stdout + r.stderr
    assert all((tmp_path / f"ok_rank{k}").exists() for k in range(world)), r.stdout + r.stderr


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
    # per-rank diagnostics of the N > 1 line: one entry per rank, each rank pinned to its own share of the host's CPUs
    assert [e["rank"] for e in d["per_rank"]] == [0, 1] and all(e["ms_per_step"] >= 0 and e["host_threads"] >= 1 for e in d["per_rank"])
    assert d["host"]["cpus"] >= 1
    r1 = subprocess.run([sys.executable, bench, "--steps", "1", "--warmup", "0", "--dry-run"], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1
    # the BARE form `python bench.py --gpus 2` (no launcher, no RANK in the environment: how the driver runs --gpus 1): bench.py re-executes
    # itself through torch.distributed.run and still prints exactly one line, from rank 0
    bare_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r2 = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"], env=bare_env, capture_output=True,
                        text=True, timeout=300)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    l2 = [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
    assert len(l2) == 1 and json.loads(l2[0])["n_gpus"] == 2 and [e["rank"] for e in json.loads(l2[0])["per_rank"]] == [0, 1]


def test_checksum_is_position_sensitive():
    import lxt_amd.dist as D
    a = torch.arange(10000, dtype=torch.float32).to(torch.bfloat16)
    assert D.checksum(a) == D.checksum(a.clone()) and D.checksum(a) != D.checksum(a.flip(0))
    b = a.clone()
    b[[3, 4]] = b[[4, 3]]
    assert D.checksum(a) != D.checksum(b)                                   # two words swapped: same multiset of words
    assert D.checksum_list([a, b]) != D.checksum_list([b, a])               # the right bytes in the wrong tensor of the list
    # long-range permutations: two words exactly 65521 apart, two whole chunks swapped, an odd storage offset
    c = torch.arange(200000, dtype=torch.int32).to(torch.int16)
    d = c.clone()
    d[[7, 7 + 65521]] = d[[7 + 65521, 7]]
    assert D.checksum(c) != D.checksum(d) and D.checksum(c, rows_per_pass=1) == D.checksum(c)
    e = torch.arange(65521 * 6, dtype=torch.int32).to(torch.int16).view(6, 65521)
    f = e.clone()
    f[[1, 5]] = f[[5, 1]]
    assert D.checksum(e) != D.checksum(f) and D.checksum(e, rows_per_pass=2) != D.checksum(f, rows_per_pass=2)
    odd = torch.arange(64, dtype=torch.uint8)[1:33]
    assert D.checksum(odd) == D.checksum(odd.clone())
    assert D.check_replicas([]) == [0]
    assert D.checksum(torch.tensor([1, 2, 3], dtype=torch.uint8)) == D.checksum(torch.tensor([1, 2, 3, 0], dtype=torch.uint8))
