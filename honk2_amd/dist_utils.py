"""Data-parallel sharding helpers: one process per GPU, clips split contiguously, logits all-gathered (RCCL).

The reference's only multi-GPU mechanism is ``torch.nn.DataParallel`` (``run/test.py:69-70``): scatter the batch,
run replicas, gather outputs on device 0.  Clips are independent on this path (BN is in eval mode), so here each
rank owns a contiguous slice and the single collective is an all-gather of the (B/G, n_labels) logits.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced split: the first n % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _flag(name):
    """An on/off environment knob: unset, empty and "0" are off."""
    return os.environ.get(name, "0") not in ("", "0")


def forced():
    """KWS_FORCE_DIST=1 under torchrun: take the distributed code path (process group, collectives, reductions) even at
    WORLD_SIZE 1 -- how a one-GPU box executes the RCCL calls the multi-GPU run makes (tests/test_gpu_parity.py)."""
    return _flag("KWS_FORCE_DIST") and "MASTER_ADDR" in os.environ


def rehearsal_backend():
    """KWS_BENCH_BACKEND=gloo: rehearse the N > 1 path with several ranks on a box with fewer GPUs (tests only; the real run is RCCL)."""
    return os.environ.get("KWS_BENCH_BACKEND") or None


def local_device_index():
    """The GPU this rank computes on: LOCAL_RANK, or 0 for every rank under the rehearsal knob KWS_BENCH_ONE_DEVICE=1."""
    return 0 if _flag("KWS_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))


def active():
    """True when results have to be combined over a process group (more than one rank, or a forced one-rank group)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    n = int(os.environ.get("WORLD_SIZE", "1"))
    if dist.is_initialized() or (n <= 1 and not forced()):
        return world()
    if backend is None:
        backend = rehearsal_backend() or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        local = local_device_index()
        if local >= torch.cuda.device_count():
            raise RuntimeError(f"rank {os.environ.get('RANK', '0')}: local rank {local} but only {torch.cuda.device_count()} GPU(s) here (one GPU per rank)")
        torch.cuda.set_device(local)                                     # one GPU per rank, before the communicator exists
    dist.init_process_group(backend=backend)
    return world()


def all_gather_rows(local, counts=None):
    """Concatenate per-rank row blocks in rank order.  `counts` (rows per rank) allows ragged shards."""
    rank, n = world()
    if n == 1:
        return local
    if counts is None:
        out = torch.empty((n * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    width = max(counts)
    padded = local
    if local.shape[0] < width:
        pad = torch.zeros((width - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    parts = [torch.empty_like(padded) for _ in range(n)]
    dist.all_gather(parts, padded.contiguous())
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)
