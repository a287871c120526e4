"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

  * approval sets / signatures are independent units -> plain index sharding, no collective
  * the MSM shards the (point, scalar) arrays by index; every rank reduces its shard to ONE
    affine point, the partials are ALL-GATHERed (world x 68 bytes -- RCCL cannot add curve
    points, and at this size the xGMI ring is latency-, not bandwidth-bound) and every rank
    adds the `world` partials locally with a unit-scalar MSM.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous, balanced index range of `rank` (the first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def msm_sharded(local_msm, combine_msm, points_shard, scalars_shard, group=None, device=None):
    """local_msm(points, scalars) -> (uint64[8] affine, is_inf) on this rank's shard;
    combine_msm(points[k,8], scalars[k,4]) the same function used on the gathered partials.
    Returns the full result on every rank."""
    import torch
    import torch.distributed as dist
    out, inf = local_msm(points_shard, scalars_shard)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return out, inf
    mine = torch.zeros(9, dtype=torch.int64, device=device)
    mine[:8] = torch.from_numpy(np.asarray(out, dtype=np.uint64).view(np.int64).copy()).to(mine.device)
    mine[8] = 1 if inf else 0
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    parts = torch.stack(gathered).cpu().numpy()
    pts = parts[:, :8].copy().view(np.uint64)
    pts[parts[:, 8] != 0] = 0            # (0, 0) is the point at infinity in the gnark layout
    ones = np.zeros((world, 4), dtype=np.uint64)
    ones[:, 0] = 1
    return combine_msm(pts, ones)
