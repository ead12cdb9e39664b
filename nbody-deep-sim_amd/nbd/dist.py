"""Range partition + per-step position exchange for the direct-force path.

The reference has no distributed code at all; this layer is the build's addition. One process
per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
Rank p owns the contiguous particle range [lo, hi); per force evaluation every rank needs all
packed sources {x,y,z,m}, so the only collective is ONE all-gather of float4[n/P] per step.
Targets are independent given the sources: no reduction, no halo, no other exchange.

Nothing here touches a kernel, so it runs unchanged on CPU tensors under gloo.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class RangePartition:
    """Contiguous, balanced ranges: the first n % P ranks get one extra body."""

    def __init__(self, n: int, world_size: int, rank: int):
        if world_size < 1 or not (0 <= rank < world_size) or n < 0:
            raise ValueError(f"bad partition n={n} world_size={world_size} rank={rank}")
        self.n, self.world_size, self.rank = int(n), int(world_size), int(rank)
        base, extra = divmod(self.n, self.world_size)
        self.counts = [base + (1 if r < extra else 0) for r in range(self.world_size)]
        self.offsets = [0]
        for c in self.counts[:-1]:
            self.offsets.append(self.offsets[-1] + c)
        self.lo = self.offsets[rank]
        self.hi = self.lo + self.counts[rank]
        self.n_local = self.counts[rank]
        self.uniform = extra == 0
        self.max_count = max(self.counts) if self.counts else 0


def allgather_rows(local: torch.Tensor, part: RangePartition, out: torch.Tensor,
                   group=None, scratch: torch.Tensor | None = None) -> torch.Tensor:
    """Gather every rank's `local` rows (n_local, C) into out[:n] in global particle order.

    Equal shards: one all_gather_into_tensor straight into `out` (a single ncclAllGather).
    Ragged shards: each rank pads to max_count, one gather into `scratch`, then the valid
    slices are compacted into `out` -- still exactly one collective per call.
    """
    if local.shape[0] != part.n_local:
        raise ValueError(f"local has {local.shape[0]} rows, partition says {part.n_local}")
    cols = local.shape[1]
    if part.world_size == 1:
        out[:part.n].copy_(local)
        return out
    if part.uniform:
        dist.all_gather_into_tensor(out[:part.n], local.contiguous(), group=group)
        return out
    m = part.max_count
    if scratch is None or scratch.shape[0] < part.world_size * m:
        scratch = torch.empty((part.world_size * m, cols), dtype=local.dtype, device=local.device)
    padded = torch.zeros((m, cols), dtype=local.dtype, device=local.device)
    padded[:part.n_local].copy_(local)
    dist.all_gather_into_tensor(scratch[:part.world_size * m], padded, group=group)
    for r in range(part.world_size):
        c = part.counts[r]
        out[part.offsets[r]:part.offsets[r] + c].copy_(scratch[r * m:r * m + c])
    return out


def group_info(group=None) -> tuple[int, int]:
    """(world_size, rank) of `group`, or (1, 0) when torch.distributed is not initialised."""
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)
