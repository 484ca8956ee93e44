"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).

The path shards without any data-path collective: every (pixel, sample) is independent and a sample's random numbers depend
only on (seed, pixel, sample index), so rank r simply renders its own range of sample indices for every pixel.  The single
exchange step is the sum of the per-rank films (Scene::m_Pixels: rgb sums + sample counts) -- one all-reduce of
width*height*4 fp32 values (10.2 MB at 800x800, 132.7 MB at 4K), which also merges the sample counts.
"""
from __future__ import annotations


def first_sample(step: int, rank: int, world: int, spp_per_rank: int, base: int = 0) -> int:
    """Sample-index range of (step, rank): disjoint across ranks and steps, contiguous within one step."""
    return base + (step * world + rank) * spp_per_rank


def sample_share(step: int, rank: int, world: int, job_spp: int, base: int = 0):
    """Strong scaling by samples: (first sample index, number of samples) of `rank` in step `step` when every step is ONE job of `job_spp`
    samples per pixel divided over `world` ranks -- contiguous, disjoint, covering [step * job_spp, (step + 1) * job_spp) exactly; shares
    differ by at most one sample when world does not divide job_spp."""
    lo, hi = job_spp * rank // world, job_spp * (rank + 1) // world
    return base + step * job_spp + lo, hi - lo


def all_reduce_film(accum, group=None):
    """In-place sum of the film accumulator over all ranks (RCCL ring all-reduce over xGMI; per-link bound ~153 GB/s)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return accum


def tile_shard(rank: int, world: int):
    """(tile_mod, tile_rem) of the interleaved pixel-tile partition (BASELINE.json configs[3] wording): rank r renders every 8x8 tile t
    with t % world == r, for ALL samples; the per-rank films are disjoint and the same all-reduce sums them into the full image."""
    return world, rank
