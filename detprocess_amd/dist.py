"""Multi-GPU sharding: one process per GPU, contiguous event ranges, no
data-path collective; the only exchange is the final all-gather of the feature
matrix over RCCL (backend "nccl" on ROCm) -- SURVEY.md section 8e.  The
reference's counterpart is Pool.starmap over series lists followed by
pd.concat (features.py:405-420)."""

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Events [lo, hi) owned by `rank`: contiguous, sizes differ by at most 1."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_features(local, total, rank, world, group=None):
    """All-gather the [B_local, F] feature matrices into [total, F] (every rank).

    Shards may differ by one row, so rows are padded to the largest shard for
    the collective and trimmed afterwards.
    """
    if world == 1:
        return local
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    F = local.shape[1]
    pad = torch.zeros((mx, F), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx, F), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
