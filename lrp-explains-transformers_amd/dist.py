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


SINGLE_RANK_COLLECTIVES = False      # module attribute (tests): True sends a ONE-rank job through the collectives too instead of short-cutting them
                                     # (RCCL on the one GPU a test box has: the same calls, buffers and dtypes as the 8-rank job)


def _collectives_off():
    return not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not SINGLE_RANK_COLLECTIVES)


BCAST_CHUNK = 1 << 30     # elements per broadcast call (module attribute: tests lower it)


def broadcast_weights(tensors, src=0):
    """in-place broadcast of a list of (already allocated) tensors from rank `src`"""
    if _collectives_off():
        return
    for t in tensors:
        if t.is_contiguous() and t.numel() > BCAST_CHUNK:
            # pieces of <= 2^30 elements (2 GiB of bf16): the flat weight buffer of an 8B model is 8.0e9 elements -- beyond INT32_MAX, and this path
            # has never run on more than one rank (no multi-GPU box was available in any round); eight 2-GiB broadcasts cost nothing against one
            # 16-GiB one (per-link bound either way) and keep every count a collective library has been exercised with
            flat = t.view(-1)
            for i in range(0, flat.numel(), BCAST_CHUNK):
                dist.broadcast(flat[i: i + BCAST_CHUNK], src=src)
        else:
            dist.broadcast(t, src=src)


_CK_MOD = (1 << 61) - 1


_CK_ROW = 65521


def checksum(t, rows_per_pass=256):
    """exact, POSITION-SENSITIVE integer checksum of a tensor's BYTES.  The 16-bit words (taken as unsigned) are laid out in rows of 65521; word j
    of a row is weighted by (1 + j) and row r's sum by (1 + r): no two positions of the buffer carry the same pair of weights, so neither two
    words a multiple of 65521 apart nor two whole blocks can be swapped unnoticed (the round-5 weight (1 + i mod 65521) alone repeated).
    Row sums are formed on the device in int64 (< 2^16 * 2^16 * 2^16), `rows_per_pass` rows = 16.8 M words at a time (temporaries: 2 x 134 MB of
    int64, not 4 x a 16-GB weight buffer; one host sync per pass), and combined in Python integers modulo 2^61 - 1 -> non-negative int.  Tensors
    whose storage offset or byte count is odd are copied / zero-extended first.  Independent of device and dtype."""
    b = t.detach().contiguous().view(-1).view(torch.uint8)
    if b.numel() % 2:
        b = torch.cat([b, b.new_zeros(1)])
    if b.data_ptr() % 2:                                       # (view(int16) needs 2-byte alignment: a uint8 slice may start on an odd byte)
        b = b.clone()
    v = b.view(torch.int16)
    per = rows_per_pass * _CK_ROW
    wgt = torch.arange(1, _CK_ROW + 1, device=v.device, dtype=torch.int64)
    tot, row0 = 0, 0
    for i in range(0, v.numel(), per):
        w = v[i: i + per].to(torch.int64).bitwise_and_(0xFFFF)
        if w.numel() % _CK_ROW:
            w = torch.cat([w, w.new_zeros(_CK_ROW - w.numel() % _CK_ROW)])
        sums = w.view(-1, _CK_ROW).mul_(wgt).sum(1).tolist()
        for r, s_ in enumerate(sums):
            tot = (tot + s_ * (row0 + r + 1)) % _CK_MOD
        row0 += len(sums)
    return tot


def checksum_list(tensors):
    """the tensors of a list in ORDER: each tensor's checksum enters with its position (a broadcast that filled the wrong tensor of the list
    changes it); 0 for an empty list"""
    tot = 0
    for k, t in enumerate(tensors):
        tot = (tot * 1000003 + checksum(t) + k + 1) % _CK_MOD
    return tot


def check_replicas(tensors):
    """after broadcast_weights: every rank's copy of `tensors` carries rank 0's bytes -- one all-gather of the per-rank checksums,
    AssertionError on every rank if any replica differs.  Returns the checksum list (one entry per rank)."""
    mine = checksum_list(tensors)                            # non-negative, < 2^61
    if _collectives_off() or not tensors:
        return [mine]
    dev = tensors[0].device
    # (an int64 does not survive a float reduction; split into 31-bit halves carried as int64 through all_gather)
    loc = torch.tensor([mine & 0x7FFFFFFF, (mine >> 31) & 0x7FFFFFFF], dtype=torch.int64, device=dev)
    allc = torch.empty(dist.get_world_size() * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, loc)
    sums = [int(a) | (int(b) << 31) for a, b in allc.view(-1, 2).tolist()]
    assert all(x == sums[0] for x in sums), f"weight replicas differ after the broadcast: per-rank checksums {sums}"
    return sums


def check_gather_order(n_total, S, device, dtype=torch.float32):
    """gather_relevance puts row p of the job at row p on every rank: each rank tags its shard's rows with their GLOBAL prompt index
    (and its own rank in column 1), gathers, and checks the result against arange -- the same code path as the job's own all-gather"""
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    lo, hi = shard_range(n_total, rank, world)
    tag = torch.zeros(hi - lo, S, device=device, dtype=dtype)
    tag[:, 0] = torch.arange(lo, hi, device=device, dtype=dtype)
    if S > 1:
        tag[:, 1] = float(rank)
    out = gather_relevance(tag, n_total)
    assert out.shape == (n_total, S), out.shape
    assert torch.equal(out[:, 0].cpu(), torch.arange(n_total, dtype=dtype)), "gathered rows are not in global prompt order"
    if S > 1:
        owner = torch.cat([torch.full((b - a,), float(r)) for r in range(world) for a, b in [shard_range(n_total, r, world)]]).to(dtype)
        assert torch.equal(out[:, 1].cpu(), owner), "gathered rows do not come from the ranks that own them"
    return True


def gather_relevance(R_local, n_total):
    """R_local [n_local, S] -> [n_total, S] on every rank, rows in global prompt order."""
    if _collectives_off():
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
