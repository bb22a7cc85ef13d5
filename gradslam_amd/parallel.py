"""One-sequence-per-GPU sharding.

Every function on the path treats batch elements independently and a single sequence is strictly
serial in time, so the only parallel axis is the batch of sequences: rank r owns sequences
`shard_indices(B, world, r)`, runs the whole SLAM loop on its own GPU with NO communication, and one
collective at the end gathers the poses (RCCL over xGMI on GPUs; gloo in the CPU tests).  The payload
is tiny (B*L*64 bytes), so ring-vs-direct does not matter; maps stay resident on their GPU and are
gathered only on request: all four attributes, per-sequence counts, reassembled in batch order into the
reference's zero-padded (B, max N_b, C) layout (structures/pointclouds.py:960-995).
"""
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

__all__ = ["init_from_env", "shard_indices", "gather_poses", "gather_ragged", "gather_maps", "maps_to_pointclouds", "run_sharded"]


def init_from_env(backend: Optional[str] = None):
    """(rank, world, local_rank) from torchrun's environment; initialises the process group when
    WORLD_SIZE > 1 (backend "nccl" == RCCL on ROCm)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # whatever the backend: the kernels run on the CURRENT device's stream (_native.require_hip), so a rank
        # selects its GPU here, before anything allocates (rehearsals may put several ranks on one card)
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("GS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
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
    """all_gatherv of one (N_r, C) tensor per rank: sizes first, then padded rows."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [rows]
    world = dist.get_world_size()
    dev = rows.device
    xdev = torch.device("cpu") if dist.get_backend() == "gloo" else dev
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=xdev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes + [1])
    pad = torch.zeros((cap, rows.shape[1]), dtype=rows.dtype, device=xdev)
    pad[: rows.shape[0]] = rows.to(xdev)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [out[r][: sizes[r]].to(dev) for r in range(world)]


_ATTRS = (("points", "points_list"), ("normals", "normals_list"), ("colors", "colors_list"), ("features", "features_list"))


def gather_maps(local_map, batch: int, device=None) -> Dict[str, object]:
    """Every rank's per-sequence maps -> the whole batch on every rank, in batch order, in the reference's padded
    layout (structures/pointclouds.py:960-995): {"counts": [N_0..N_{B-1}], "points" / "normals" / "colors" /
    "features": (B, max N_b, C) zero-padded, or None for an attribute no rank holds}.

    `local_map` is this rank's Pointclouds (its `*_list` attributes are read; None or empty for a rank without
    sequences).  Two collectives: the per-sequence counts and attribute widths, then ONE all_gather of the rows
    with the attributes packed side by side (40 B per point for a PointFusion map), padded to the global maximum."""
    lists = {}
    for name, attr in _ATTRS:
        v = getattr(local_map, attr, None) if local_map is not None else None
        lists[name] = list(v) if v is not None and len(v) > 0 else None
    n_local = len(lists["points"]) if lists["points"] is not None else 0
    if device is None:
        device = lists["points"][0].device if n_local else torch.device("cpu")
    widths = [lists[k][0].shape[1] if lists[k] is not None else 0 for k, _ in _ATTRS]
    counts = [int(x.shape[0]) for x in lists["points"]] if n_local else []
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    shards = [shard_indices(batch, world, r) for r in range(world)]
    assert n_local == len(shards[rank]), "rank {} holds {} sequences, its shard has {}".format(rank, n_local, len(shards[rank]))
    cap = max(len(s) for s in shards)
    xdev = torch.device("cpu") if (distributed and dist.get_backend() == "gloo") else device
    # (1) counts + attribute widths of every rank
    meta = torch.zeros(cap + 4, dtype=torch.int64, device=xdev)
    meta[:n_local] = torch.tensor(counts, dtype=torch.int64)
    meta[cap:] = torch.tensor(widths, dtype=torch.int64)
    if distributed:
        metas = [torch.empty_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta)
    else:
        metas = [meta]
    metas = [m.tolist() for m in metas]
    all_counts = [metas[r][i] for r in range(world) for i in range(len(shards[r]))]
    # a rank without sequences knows no widths: take them from the ranks that have some
    gw = [max(metas[r][cap + a] for r in range(world)) for a in range(4)]
    for r in range(world):
        for a in range(4):
            assert metas[r][cap + a] in (0, gw[a]) or not shards[r], "attribute widths differ between ranks"
    out = {"counts": all_counts, "points": None, "normals": None, "colors": None, "features": None}
    ctot, nmax = sum(gw), max(all_counts + [0])
    if ctot == 0 or batch == 0:
        return out
    # (2) one padded block per rank: (cap, nmax, ctot), attributes side by side
    block = torch.zeros((cap, max(nmax, 1), ctot), dtype=torch.float32, device=xdev)
    for i in range(n_local):
        col = 0
        for a, (name, _) in enumerate(_ATTRS):
            if gw[a] and lists[name] is not None:
                block[i, : counts[i], col: col + gw[a]] = lists[name][i].detach().to(xdev)
            col += gw[a]
    if distributed:
        blocks = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(blocks, block)
    else:
        blocks = [block]
    whole = torch.cat([blocks[r][: len(shards[r])] for r in range(world)], 0)[:, :nmax].to(device)
    col = 0
    for a, (name, _) in enumerate(_ATTRS):
        if gw[a]:
            out[name] = whole[:, :, col: col + gw[a]].contiguous()
        col += gw[a]
    return out


def maps_to_pointclouds(gathered: Dict[str, object]):
    """The gathered batch as a Pointclouds (lists cut from the padded arrays at the per-sequence counts)."""
    from .structures.pointclouds import Pointclouds

    n = gathered["counts"]
    cut = lambda x: None if x is None else [x[b, : n[b]] for b in range(len(n))]
    if gathered["points"] is None:
        return Pointclouds()
    return Pointclouds(points=cut(gathered["points"]), normals=cut(gathered["normals"]), colors=cut(gathered["colors"]),
                       features=cut(gathered["features"]))


def run_sharded(slam_fn, colors, depths, intrinsics, poses, *, gather_maps: bool = False):
    """Run `slam_fn(colors_r, depths_r, intrinsics_r, poses_r) -> (Pointclouds, poses (B_r,L,4,4))` on this rank's
    shard of the batch and gather the poses.  Returns (local map, all poses) or, with gather_maps=True,
    (local map, all poses, gathered maps as returned by `gather_maps`).  A rank whose shard is empty (B < world)
    skips `slam_fn` and only takes part in the collectives."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    B, L = colors.shape[0], colors.shape[1]
    mine = shard_indices(B, world, rank)
    sel = lambda x: None if x is None else x[mine]
    if mine:
        local_map, local_poses = slam_fn(sel(colors), sel(depths), sel(intrinsics), sel(poses))
    else:
        local_map, local_poses = None, torch.zeros((0, L, 4, 4), dtype=torch.float32, device=colors.device)
    all_poses = gather_poses(local_poses, B)
    if not gather_maps:
        return local_map, all_poses
    return local_map, all_poses, globals()["gather_maps"](local_map, B, device=colors.device)
