"""One-sequence-per-GPU sharding.

Every function on the path treats batch elements independently and a single sequence is strictly
serial in time, so the only parallel axis is the batch of sequences: rank r owns sequences
`shard_indices(B, world, r)`, runs the whole SLAM loop on its own GPU with NO communication, and one
collective at the end gathers the poses (RCCL over xGMI on GPUs; gloo in the CPU tests).  The payload
is tiny (B*L*64 bytes), so ring-vs-direct does not matter; maps stay resident on their GPU and are
gathered only on request (padded to the global maximum, like the reference's padded layout).
"""
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

__all__ = ["init_from_env", "shard_indices", "gather_poses", "gather_ragged", "run_sharded"]


def init_from_env(backend: Optional[str] = None):
    """(rank, world, local_rank) from torchrun's environment; initialises the process group when
    WORLD_SIZE > 1 (backend "nccl" == RCCL on ROCm)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("GS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_indices(batch: int, world: int, rank: int) -> List[int]:
    """Contiguous block partition of `batch` sequences over `world` ranks (first ranks get the
    remainder); deterministic and order preserving so the gather reassembles the batch as-is."""
    base, rem = divmod(batch, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def gather_poses(local_poses: torch.Tensor, batch: int) -> torch.Tensor:
    """local (B_r, L, 4, 4) on every rank -> (B, L, 4, 4) on every rank, in batch order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_poses
    world = dist.get_world_size()
    L = local_poses.shape[1]
    dev = local_poses.device
    # gloo (CPU rehearsal of the multi-rank path) gathers host tensors; RCCL gathers in HBM over xGMI
    xdev = torch.device("cpu") if dist.get_backend() == "gloo" else dev
    cap = max(len(shard_indices(batch, world, r)) for r in range(world))
    pad = torch.zeros((cap, L, 4, 4), dtype=local_poses.dtype, device=xdev)
    pad[: local_poses.shape[0]] = local_poses.to(xdev)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([out[r][: len(shard_indices(batch, world, r))] for r in range(world)], 0).to(dev)


def gather_ragged(rows: torch.Tensor) -> List[torch.Tensor]:
    """all_gatherv of one (N_r, C) tensor per rank (map attributes): sizes first, then padded rows."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [rows]
    world = dist.get_world_size()
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes + [1])
    pad = torch.zeros((cap, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [out[r][: sizes[r]] for r in range(world)]


def run_sharded(slam_fn, colors, depths, intrinsics, poses, *, gather_maps: bool = False):
    """Run `slam_fn(colors_r, depths_r, intrinsics_r, poses_r) -> (Pointclouds-like, poses (B_r,L,4,4))`
    on this rank's shard of the batch and gather the poses.  Returns (local map, all poses[, maps])."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    B = colors.shape[0]
    mine = shard_indices(B, world, rank)
    sel = lambda x: None if x is None else x[mine]
    local_map, local_poses = slam_fn(sel(colors), sel(depths), sel(intrinsics), sel(poses))
    all_poses = gather_poses(local_poses, B)
    if not gather_maps:
        return local_map, all_poses
    pts = torch.cat(local_map.points_list, 0) if len(mine) else colors.new_zeros((0, 3))
    return local_map, all_poses, gather_ragged(pts)
