"""Multi-GPU sharding of one frame: interleaved tiles, scene replicated, framebuffer
gathered with torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU).

The reference's only parallelism is an OpenMP loop over pixels
(reference src/raytrace.cpp:92-96); here every rank (one process per GPU) renders the
tiles it owns into a packed shard buffer, the shards travel to rank 0 in ONE collective,
and rank 0 de-interleaves them into the row-major frame.  Pixels never depend on the
partition, so the frame is byte-identical for any world size.
"""
from __future__ import annotations

import numpy as np

from .api import RT_TILE, shard_bytes


def tile_owner(tx: int, ty: int, world: int) -> int:
    """Rank owning tile (tx, ty): the rule of rt_shard_tiles / rt_render_shard."""
    return (tx + 3 * ty) % world


def tile_list(width: int, height: int, rank: int, world: int):
    """Tiles (tx, ty) of `rank` in the order its shard buffer stores them (row-major)."""
    ntx, nty = -(-width // RT_TILE), -(-height // RT_TILE)
    return [(tx, ty) for ty in range(nty) for tx in range(ntx) if tile_owner(tx, ty, world) == rank]


def pack_shard_host(frame: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Host mirror of what rt_render_shard writes: the rank's tiles of an [H,W,3] uint8
    frame, tile-major, RT_TILE*RT_TILE*3 bytes per tile (pixels outside the film zero),
    padded to the common shard size."""
    h, w, _ = frame.shape
    out = np.zeros(shard_bytes(w, h, world), np.uint8)
    tiles = tile_list(w, h, rank, world)
    view = out[: len(tiles) * RT_TILE * RT_TILE * 3].reshape(len(tiles), RT_TILE, RT_TILE, 3)
    for i, (tx, ty) in enumerate(tiles):
        blk = frame[ty * RT_TILE:(ty + 1) * RT_TILE, tx * RT_TILE:(tx + 1) * RT_TILE]
        view[i, : blk.shape[0], : blk.shape[1]] = blk
    return out


def assemble_host(all_shards: np.ndarray, world: int, width: int, height: int) -> np.ndarray:
    """Host mirror of rt_assemble_shards: world consecutive shard buffers -> [H,W,3]."""
    sb = shard_bytes(width, height, world)
    all_shards = np.asarray(all_shards, np.uint8).reshape(world, sb)
    frame = np.zeros((height, width, 3), np.uint8)
    for r in range(world):
        tiles = tile_list(width, height, r, world)
        view = all_shards[r, : len(tiles) * RT_TILE * RT_TILE * 3].reshape(len(tiles), RT_TILE, RT_TILE, 3)
        for i, (tx, ty) in enumerate(tiles):
            y0, x0 = ty * RT_TILE, tx * RT_TILE
            hh, ww = min(RT_TILE, height - y0), min(RT_TILE, width - x0)
            frame[y0:y0 + hh, x0:x0 + ww] = view[i, :hh, :ww]
    return frame


def gather_shards(shard, rank: int, world: int, dst: int = 0):
    """ONE collective per frame: every rank's packed shard (a torch uint8 tensor, all the
    same length) to rank `dst`.  Returns the concatenated [world * n] tensor on dst, else None."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return shard
    if rank == dst:
        out = torch.empty(world * shard.numel(), dtype=shard.dtype, device=shard.device)
        dist.gather(shard, list(out.chunk(world)), dst=dst)
        return out
    dist.gather(shard, None, dst=dst)
    return None


def render_frame_distributed(tracer, rank: int, world: int, device):
    """Render this rank's shard on its GPU, gather, assemble on rank 0.
    -> (frame tensor [H,W,3] uint8 on rank 0 else None, RtStats of this rank).
    When the tracer runs on torch's current stream (Tracer(stream=torch.cuda.current_stream().cuda_stream)), render,
    gather and assemble are ordered by that stream and the frame costs ONE host synchronisation (the ray counters at
    the end of the render); the frame tensor is then complete in stream order, not on return."""
    import torch

    s = tracer.scene
    sb = shard_bytes(s.width, s.height, world)
    shard = torch.empty(sb, dtype=torch.uint8, device=device)
    stats = tracer.render_shard(rank, world, shard.data_ptr())
    allsh = gather_shards(shard, rank, world)
    if rank != 0:
        return None, stats
    frame = torch.empty((s.height, s.width, 3), dtype=torch.uint8, device=device)
    if getattr(tracer, "stream", None) != torch.cuda.current_stream(device).cuda_stream:
        torch.cuda.current_stream(device).synchronize()      # the gather ran on torch's stream, the tracer has its own
    tracer.assemble_shards(allsh.data_ptr(), world, frame.data_ptr())
    return frame, stats
