"""Reach sharding across the GPUs of a node and the one collective of the path.

Reaches / ensemble members are independent: each rank owns a contiguous
block of global reach indices and steps it without talking to anyone.  The only exchange is the
gather of the boundary hydrographs [levels, 4, B_local] over RCCL (backend "nccl" on ROCm) or gloo (CPU tests): to one
root rank (what north_star asks for: every other rank sends its block straight to the root, one xGMI hop, 1/world of the bytes
an all_gather moves through every rank) or, with root=None, to every rank."""
import torch
import torch.distributed as dist


def reach_block(rank: int, world: int, per_rank: int):
    """Weak-scaling layout used by bench.py: rank r owns [r*per_rank, (r+1)*per_rank)."""
    return rank * per_rank, per_rank


def split_reaches(total: int, rank: int, world: int):
    """Strong-scaling layout: `total` reaches in `world` contiguous blocks, remainder to the first ranks."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def gather_hydrographs(local: torch.Tensor, world: int, root=None):
    """local [levels, 4, B] on every rank -> [levels, 4, world*B] in global reach order: on every rank (root=None, an
    all_gather) or on rank `root` alone (a gather: the other ranks return None)."""
    if world == 1:
        return local
    local = local.contiguous()
    if root is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(world * local.shape[0], *local.shape[1:]), local)
    else:
        me = dist.get_rank()
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device) if me == root else None
        dist.gather(local, list(out.unbind(0)) if me == root else None, dst=root)
        if me != root:
            return None
    return out.permute(1, 2, 0, 3).reshape(local.shape[0], local.shape[1], world * local.shape[2])


def gather_hydrographs_split(local: torch.Tensor, total: int, world: int, root=None):
    """The same gather for the strong-scaling layout (split_reaches): the blocks differ by at most one reach, the
    shorter ones travel padded by a column (all_gather wants equal shapes) that is dropped again on arrival.
    local [levels, 4, count(rank)] -> [levels, 4, total] in global reach order (all ranks, or `root` alone: see gather_hydrographs)."""
    if world == 1:
        return local
    counts = [split_reaches(total, r, world)[1] for r in range(world)]
    widest = max(counts)
    if local.shape[2] != widest:
        padded = torch.zeros(local.shape[:2] + (widest,), dtype=local.dtype, device=local.device)
        padded[:, :, :local.shape[2]] = local
        local = padded
    out = gather_hydrographs(local, world, root)
    if out is None or min(counts) == widest:
        return out
    keep = torch.cat([torch.arange(r * widest, r * widest + c, device=out.device) for r, c in enumerate(counts)])
    return out.index_select(2, keep)
