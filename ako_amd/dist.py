"""Multi-GPU harness of the transform path: one process per GPU, images sharded, no data-path collective.

The path shards by independent images (the reference's unit of independence is the tile /
image: library/encode.c:115-205): rank r of W transforms images r, r + W, r + 2W ... of a batch, or its
own per-rank batch in the weak-scaling benchmark.  `torch.distributed` (RCCL on GPUs, gloo on CPU) is
used ONLY for: the barrier around the timed region, the max-over-ranks of the elapsed time, and the
gather of per-image checksums onto rank 0.  Nothing here touches the GPU by itself, so it is testable
with gloo on CPU (tests/test_dist_gloo.py).
"""
from __future__ import annotations

import os
import time
from typing import Callable, List, Sequence


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str):
    import torch.distributed as dist

    _, _, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    return world


def shard_images(n_images: int, rank: int, world: int) -> List[int]:
    """Strong-scaling split: image i of a fixed batch goes to rank i mod world."""
    return list(range(rank, n_images, world))


def image_seeds(rank: int, per_rank: int, base: int = 0x9E3779B9) -> List[int]:
    """Weak-scaling split: every rank owns `per_rank` images, image j of rank r is seeded
    base + r * per_rank + j -- so N ranks together process exactly the N * per_rank distinct images a
    single rank would have been given in turn (BASELINE configs[3]: image i seeded 0x9E3779B9 + i)."""
    return [(base + rank * per_rank + j) & 0xFFFFFFFF for j in range(per_rank)]


def tile_band(image_h: int, tiles_dimension: int, rank: int, world: int):
    """Tile sharding of ONE tiled image (BASELINE configs[4]): rank r takes a contiguous band of tile
    rows, (y0, rows).  Tiles are independent and stored in raster order (library/encode.c:115-204), so a
    band of whole tile rows is itself a valid image whose stream is exactly that slice of the full
    image's stream: concatenating the ranks' streams in rank order gives the stream of the whole image.
    No collective is involved."""
    assert tiles_dimension > 0
    tile_rows = (image_h + tiles_dimension - 1) // tiles_dimension
    per = (tile_rows + world - 1) // world
    first = min(rank * per, tile_rows)
    last = min(first + per, tile_rows)
    y0 = first * tiles_dimension
    y1 = min(last * tiles_dimension, image_h)
    return y0, max(0, y1 - y0)


def barrier(sync: Callable[[], None] | None = None):
    import torch.distributed as dist

    if sync is not None:
        sync()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync is not None:
        sync()


def max_int(v: int, device=None) -> int:
    """MAX over ranks of an integer (every rank gets the same number)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def timed_steps(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None] | None = None,
                device=None, before_timed: Callable[[], None] | None = None, spread: dict | None = None) -> float:
    """W untimed steps, then exactly K steps bracketed by barrier + sync on both sides; returns the MAX
    over ranks of the elapsed seconds (every rank gets the same number).  `before_timed` runs after the
    warm-up, just before the opening barrier (e.g. to switch per-kernel event recording on).  `spread`, when
    given, receives the fastest and the slowest rank's time up to its OWN last step (before the closing
    barrier): {"min_s", "max_s"} -- a straggler shows as a gap between the two."""
    import torch
    import torch.distributed as dist

    for _ in range(warmup):
        step()
    if before_timed is not None:
        before_timed()
    barrier(sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    own = None
    if spread is not None:
        if sync is not None:
            sync()
        own = time.perf_counter() - t0
    barrier(sync)
    elapsed = time.perf_counter() - t0
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if own is not None:
            lo = torch.tensor([own], dtype=torch.float64, device=device if device is not None else "cpu")
            hi = lo.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            spread["min_s"], spread["max_s"] = float(lo.item()), float(hi.item())
    elif own is not None:
        spread["min_s"] = spread["max_s"] = own
    return elapsed


def gather_checksums(local: Sequence[int], device=None) -> List[List[int]] | None:
    """All ranks' per-image checksums on rank 0 (None elsewhere); equal counts per rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [list(local)]
    world = dist.get_world_size()
    mine = torch.tensor(list(local), dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    if dist.get_rank() != 0:
        return None
    return [o.cpu().tolist() for o in out]
