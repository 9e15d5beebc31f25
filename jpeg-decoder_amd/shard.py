"""Image sharding across GPUs and the end-of-run reduction.

Every JPEG is self-contained (own tables, own DC predictors), so a batch is partitioned
by image: rank r of G decodes its own images and no pixel or coefficient crosses xGMI.
The only collective is one reduction after the timed region: SUM over the per-rank
counters and MAX over the per-rank wall time (the slowest rank defines the job's
throughput).  The backend is whatever the process group was created with: "nccl"
(RCCL) on the GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Strong-scaling partition of a fixed list of images: item i goes to rank i mod G."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_items, world))


def shard_seeds(batch_per_gpu: int, rank: int, world: int) -> List[int]:
    """Weak-scaling partition used by the benchmark: every rank generates and decodes
    batch_per_gpu images of its own; seeds are disjoint and contiguous per rank, so the
    union over ranks is seeds 0 .. G*batch-1 whatever G is."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return [rank * batch_per_gpu + i for i in range(batch_per_gpu)]


@dataclass
class RunTotals:
    pixels: float          # decoded pixels per step, all ranks
    images: float          # images per step, all ranks
    comp_bytes: float      # compressed bytes per step, all ranks
    mismatches: float      # parity mismatches counted by the ranks (0 when not checked)
    elapsed_max: float     # slowest rank's wall time over the timed steps, seconds
    kernel_ms_max: List[float]  # per-kernel average milliseconds, max over ranks

    def megapixels_per_s(self, steps: int) -> float:
        return self.pixels * steps / self.elapsed_max / 1e6


def reduce_run(pixels: float, images: float, comp_bytes: float, mismatches: float, elapsed: float,
               kernel_ms: Sequence[float], device=None) -> RunTotals:
    """All-reduce the counters of one run.  Without an initialised process group (one
    process, one GPU) this is the identity."""
    import torch
    import torch.distributed as dist

    sums = torch.tensor([pixels, images, comp_bytes, mismatches], dtype=torch.float64, device=device)
    maxs = torch.tensor([elapsed, *kernel_ms], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(maxs, op=dist.ReduceOp.MAX)
    s = sums.tolist()
    m = maxs.tolist()
    return RunTotals(s[0], s[1], s[2], s[3], m[0], m[1:])


def gather_ranks(values: Sequence[float], device=None) -> List[List[float]]:
    """Every rank's vector of per-rank figures, in rank order: result[r][k] is values[k] of
    rank r.  One all-gather of len(values) doubles (no pixel or coefficient data).  Without
    an initialised process group this is [values]."""
    import torch
    import torch.distributed as dist

    mine = torch.tensor(list(values), dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [mine.tolist()]
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [t.tolist() for t in out]
