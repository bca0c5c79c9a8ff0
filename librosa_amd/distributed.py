"""Multi-GPU sharding of a batch of clips: one process per GPU, no collective on the data path.

Clips (leading axes) are independent (the reference asserts batch == per-item,
``tests/test_multichannel.py:96-111, 685-714``), so a batch shards by contiguous clip ranges with zero
exchange during compute.  The only collective is the optional final gather of the (small) mel
output -- RCCL over xGMI when the process group uses the ``nccl`` backend (which is RCCL on ROCm),
``gloo`` on CPU for tests.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous, balanced [begin, end) of ``n_items`` for ``rank``; the first ``n % world`` ranks get one extra."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, extra = divmod(int(n_items), world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world_size: int):
    return [shard_range(n_items, r, world_size)[1] - shard_range(n_items, r, world_size)[0] for r in range(world_size)]


def gather_shards(local, n_items: int, group=None):
    """All-gather per-rank results (clips on axis 0, possibly unequal shard sizes) into the full batch.

    ``local`` is a torch tensor holding this rank's shard.  Every rank returns the full
    ``(n_items, ...)`` tensor.  Uses one padded ``all_gather_into_tensor`` (a direct exchange over
    xGMI under RCCL; each rank's shard crosses each link once).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = shard_sizes(n_items, world)
    biggest = max(sizes)
    pad = local
    if local.shape[0] < biggest:
        pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(s == biggest for s in sizes):
        return out
    return torch.cat([out[r * biggest : r * biggest + sizes[r]] for r in range(world)], dim=0)
