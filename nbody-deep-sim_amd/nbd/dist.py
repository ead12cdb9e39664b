"""Range partition + per-step position exchange for the direct-force path.

The reference has no distributed code at all; this layer is the build's addition. One process
per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
Rank p owns the contiguous particle range [lo, hi); per force evaluation every rank needs all
packed sources {x,y,z,m}, so the only collective is ONE all-gather of float4[n/P] per step.
Targets are independent given the sources: no reduction, no halo, no other exchange. The gather is
asynchronous (RowGather.start/finish): the simulator runs the force of the rank's own bodies on
each other while the other ranks' bodies are in flight.

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


class RowGather:
    """The per-step exchange: every rank's (n_local, C) rows -> `out[:n]` in global particle order.

    Exactly one collective per exchange, asynchronous: `start()` enqueues it (on RCCL's own stream behind
    the caller's current stream) and returns at once, so kernels launched next overlap the transfer;
    `finish()` makes the current stream wait for it. All buffers are allocated here, once.

    Equal shards: one all_gather_into_tensor straight into `out`. Ragged shards (n % P != 0): every rank
    sends `max_count` rows (its own rows + zero padding: callers keep `local` that long), one gather into
    a scratch array, then one index_select compacts the valid rows into `out`.
    """

    def __init__(self, part: RangePartition, cols: int, dtype, device, group=None, collective: bool = False):
        self.part, self.group = part, group
        self.collective = collective or part.world_size > 1    # a one-rank group can still go through the collective
        self.send_rows = part.n_local if part.uniform else part.max_count
        self.scratch = self.index = None
        if part.world_size > 1 and not part.uniform:
            m = part.max_count
            self.scratch = torch.empty((part.world_size * m, cols), dtype=dtype, device=device)
            self.index = torch.cat([torch.arange(r * m, r * m + c, dtype=torch.int64)
                                    for r, c in enumerate(part.counts)]).to(device)

    def start(self, local: torch.Tensor, out: torch.Tensor):
        """`local`: at least `send_rows` contiguous rows (rows past n_local must be zero padding)."""
        part = self.part
        if local.shape[0] < self.send_rows:
            raise ValueError(f"local has {local.shape[0]} rows, the exchange sends {self.send_rows}")
        if not self.collective:
            out[:part.n].copy_(local[:part.n])
            return None
        dst = out[:part.n] if part.uniform else self.scratch
        return dist.all_gather_into_tensor(dst, local[:self.send_rows], group=self.group, async_op=True)

    def finish(self, handle, out: torch.Tensor) -> torch.Tensor:
        if handle is not None:
            handle.wait()          # device tensors: the current stream waits; host tensors (gloo): blocks
            if self.scratch is not None:
                torch.index_select(self.scratch, 0, self.index, out=out[:self.part.n])
        return out


def allgather_rows(local: torch.Tensor, part: RangePartition, out: torch.Tensor, group=None) -> torch.Tensor:
    """Blocking form for occasional use (state gathers for output, energies): allocates per call."""
    if local.shape[0] != part.n_local:
        raise ValueError(f"local has {local.shape[0]} rows, partition says {part.n_local}")
    g = RowGather(part, local.shape[1], local.dtype, local.device, group)
    if g.send_rows > local.shape[0]:
        padded = torch.zeros((g.send_rows, local.shape[1]), dtype=local.dtype, device=local.device)
        padded[:part.n_local].copy_(local)
        local = padded
    return g.finish(g.start(local.contiguous(), out), out)


def group_info(group=None) -> tuple[int, int]:
    """(world_size, rank) of `group`, or (1, 0) when torch.distributed is not initialised."""
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)
