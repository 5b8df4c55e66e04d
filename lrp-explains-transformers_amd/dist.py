"""One process per GPU: shard a batch of explanation jobs over the ranks of one node.

The path shards embarrassingly: every prompt's explanation is independent, so there
is NO collective on the data path.  RCCL (torch.distributed backend "nccl" on ROCm) over xGMI is used
for exactly two things, neither per layer:
  * broadcast_weights : rank 0's weights -> every rank, once at start-up: the engine keeps every weight in ONE flat buffer
                        (LlamaLRP.flat, forward layouts only), so this is a single collective; nothing is rebuilt afterwards
                        (the dgrad GEMMs read the stored weights: no W^T copies; LlamaLRP.build_transposes only drops the
                        fp32 parity engine's lazily cached transposes);
  * gather_relevance  : all-gather of the [n_local, S] fp32 token relevances at the end of a job.
Correctness contract: the rank-sharded relevance of prompt p equals the single-GPU relevance of p bit
for bit (same kernels, same order) -- tests/test_dist_cpu.py checks the partition/gather logic with the
gloo backend at world_size 2, tests/test_engine_gpu.py::test_llama_batch_equals_single the kernels.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).
    Returns (rank, world, local_rank).  Single-process runs need no initialisation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_range(n_items, rank, world):
    """contiguous block partition (equal S => equal work); the first n_items % world ranks get one more"""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_weights(tensors, src=0):
    """in-place broadcast of a list of (already allocated) tensors from rank `src`"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src)


def gather_relevance(R_local, n_total):
    """R_local [n_local, S] -> [n_total, S] on every rank, rows in global prompt order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return R_local
    world = dist.get_world_size()
    S = R_local.shape[1]
    n_max = -(-n_total // world)
    pad = torch.zeros(n_max, S, device=R_local.device, dtype=R_local.dtype)
    pad[: R_local.shape[0]] = R_local
    out = torch.empty(world * n_max, S, device=R_local.device, dtype=R_local.dtype)
    dist.all_gather_into_tensor(out, pad)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        rows.append(out[r * n_max: r * n_max + (hi - lo)])
    return torch.cat(rows, 0)


def explain_sharded(explain_fn, ids, batch, rank=None, world=None, gather=True):
    """Run explain_fn(ids_chunk [b,S]) -> R [b,S] over this rank's shard of `ids` [n,S] in chunks of
    `batch`, then all-gather ONCE per job.  Returns R [n,S] (global order) on every rank.
    rank / world override the process group's (gather=False: return only the local shard) -- used to check on ONE device that
    the shards of a job reproduce the un-sharded result bit for bit."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(ids.shape[0], rank, world)
    outs = [explain_fn(ids[i: min(i + batch, hi)]) for i in range(lo, hi, batch)]
    S = ids.shape[1]
    R_local = torch.cat(outs, 0) if outs else torch.zeros(0, S, dtype=torch.float32, device=ids.device)
    return gather_relevance(R_local, ids.shape[0]) if gather else R_local
