"""Multi-GPU layout of the path: reads are sharded over ranks (one process per GPU), every
alignment depends on exactly one read, so phases B and C need no exchange at all.  The single
cross-read reduction in Porechop is the adapter-set presence table of phase A -- the max
full-adapter identity per set and side over the check reads (nanopore_read.py:159,164), consumed
at porechop.py:327 -- which is all-reduced with MAX (RCCL on GPUs, gloo in the CPU tests)."""
import os

import torch
import torch.distributed as dist


def _collective_needed(group=None):
    """More than one rank -- or PC_DIST_FORCE_COLLECTIVES=1 (tests: a ONE-rank RCCL group on a single-GPU box still takes every
    call through the RCCL API: initialisation with a device, float64 MAX, int64 MIN / all_gather on device tensors)."""
    return dist.get_world_size(group) > 1 or os.environ.get("PC_DIST_FORCE_COLLECTIVES", "0") not in ("", "0")


def shard_bounds(n_items, world, rank):
    """Contiguous, order-preserving shard [lo, hi) of rank `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_bases(lengths, world, rank):
    """Contiguous, order-preserving shard [lo, hi) of rank `rank` with about 1/world of the BASES
    (the middle scan's work is proportional to read length)."""
    import numpy as np
    n = int(len(lengths))
    if world <= 1 or n == 0:
        return 0, n
    ends = np.cumsum(np.asarray(lengths, dtype=np.int64))
    total = int(ends[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(ends, total * r / world, side="left")))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.array(cuts))
    return int(cuts[rank]), int(cuts[rank + 1])


def check_read_share(check_reads, world, rank, n_total):
    """Porechop checks the FIRST `check_reads` reads of the input (porechop.py:224-273).  With the
    input sharded contiguously (shard_bounds), rank r holds global reads [lo, hi): its share of the
    check set is the overlap with [0, check_reads).  Returns how many of its leading reads to check."""
    lo, hi = shard_bounds(n_total, world, rank)
    return max(0, min(hi, check_reads) - lo)


def _host_collectives(group=None):
    """gloo moves bytes through host memory and implements only some collectives for device tensors: with it (the CPU
    tests, and functional checks of the N > 1 path on a single-GPU box) device tensors take the host route explicitly.
    RCCL ("nccl") works on the device tensors in place."""
    return dist.get_backend(group) == "gloo"


def reduce_presence(best_start, best_end, group=None):
    """MAX all-reduce of the [S] + [S] presence tables (in place on a stacked copy)."""
    table = torch.stack([best_start, best_end])
    if dist.is_available() and dist.is_initialized() and _collective_needed(group):
        if table.is_cuda and _host_collectives(group):
            host = table.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
            table = host.to(table.device)
        else:
            dist.all_reduce(table, op=dist.ReduceOp.MAX, group=group)
    return table[0], table[1]


def gather_in_order(local, group=None):
    """Concatenate per-rank 1-D/2-D tensors in rank order (= read order for contiguous shards)."""
    if not (dist.is_available() and dist.is_initialized()) or not _collective_needed(group):
        return local
    world = dist.get_world_size(group)
    if local.is_cuda and _host_collectives(group):
        return gather_in_order(local.cpu(), group).to(local.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: int(s.item())] for p, s in zip(parts, sizes)])


def _staging_device(device=None):
    """Where small host-side values must live to take part in a collective: RCCL ("nccl") only moves device tensors."""
    if dist.get_backend() == "nccl":
        return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def all_gather_ints(values, device=None):
    """Small per-rank integer vectors of EQUAL length -> int64 tensor [world, len] on the host (same on every rank)."""
    dev = _staging_device(device)
    mine = torch.as_tensor(values, dtype=torch.int64).reshape(-1).to(dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return torch.stack(parts).cpu()


def all_agree(flag, device=None):
    """True iff `flag` is true on EVERY rank (MIN all-reduce of one integer)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=_staging_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def all_gather_objects(obj):
    """Small picklable per-rank values -> list over ranks (same on every rank)."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
