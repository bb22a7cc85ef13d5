"""World-size-2 gloo tests of the one-sequence-per-GPU sharding (CPU processes stand in for ranks): the pose
gather and the full map gather -- all four attributes, ragged per-sequence sizes, uneven shards and an empty shard
-- reassembled on every rank in batch order into the reference's padded layout
(structures/pointclouds.py:960-995).  The per-rank "SLAM" here is a deterministic stand-in because the HIP kernels
need a GPU; tests/test_gpu_parity.py::test_run_sharded_real_slam_two_ranks runs the real PointFusion through the
same `run_sharded` on the GPU box."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from gradslam_amd import parallel
from gradslam_amd.structures.pointclouds import Pointclouds


def test_shard_indices_partition():
    for B in (1, 2, 5, 8, 9):
        for w in (1, 2, 3, 8):
            parts = [parallel.shard_indices(B, w, r) for r in range(w)]
            assert sum(parts, []) == list(range(B))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _seq_map(i: int, with_feats: bool = True):
    """Deterministic map of sequence i: 3 i + 1 points (sequence 3 is EMPTY), every attribute distinct."""
    n = 0 if i == 3 else 3 * i + 1
    g = torch.Generator().manual_seed(100 + i)
    mk = lambda c: torch.rand((n, c), generator=g) + float(i)
    return mk(3), mk(3), mk(3), (mk(1) if with_feats else None)


def _whole_batch(B: int, with_feats: bool = True):
    cols = list(zip(*[_seq_map(i, with_feats) for i in range(B)]))
    return Pointclouds(points=list(cols[0]), normals=list(cols[1]), colors=list(cols[2]),
                       features=list(cols[3]) if with_feats else None)


def _worker(rank, world, port, B, L, with_feats, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    colors = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1, 1).expand(B, L, 2, 2, 3).contiguous()

    def slam_fn(c, d, k, p):  # per-sequence "SLAM": pose (b, l) encodes (sequence id, frame id)
        ids = [int(x) for x in c[:, 0, 0, 0, 0]]
        poses = torch.eye(4).view(1, 1, 4, 4).repeat(len(ids), L, 1, 1)
        poses[:, :, 0, 3] = torch.tensor(ids, dtype=torch.float32).view(-1, 1)
        poses[:, :, 1, 3] = torch.arange(L, dtype=torch.float32).view(1, -1)
        cols = list(zip(*[_seq_map(i, with_feats) for i in ids]))
        return Pointclouds(points=list(cols[0]), normals=list(cols[1]), colors=list(cols[2]),
                           features=list(cols[3]) if with_feats else None), poses

    _, all_poses, maps = parallel.run_sharded(slam_fn, colors, None, None, None, gather_maps=True)
    torch.save({"poses": all_poses, "maps": maps}, os.path.join(out_dir, "r{}.pt".format(rank)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("B,with_feats", [(2, True), (5, True), (1, True), (4, False)])
def test_sharded_gather_world2(tmp_path, B, with_feats):
    """B=5: uneven shards (3 + 2) with an empty sequence (id 3); B=1: rank 1's shard is empty; B=4 without
    features: ICPSLAM's aggregate map."""
    L, world = 3, 2
    mp.spawn(_worker, args=(world, _free_port(), B, L, with_feats, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "r{}.pt".format(r))) for r in range(world)]
    want = _whole_batch(B, with_feats)
    for o in outs:  # every rank holds the whole batch, in batch order
        assert o["poses"].shape == (B, L, 4, 4)
        assert o["poses"][:, 0, 0, 3].tolist() == [float(b) for b in range(B)]
        assert o["poses"][0, :, 1, 3].tolist() == [0.0, 1.0, 2.0]
        m = o["maps"]
        assert m["counts"] == want.num_points_per_pointcloud.tolist()
        # the reference's padded layout: (B, max N_b, C), zero beyond each sequence's count
        assert torch.equal(m["points"], want.points_padded)
        assert torch.equal(m["normals"], want.normals_padded)
        assert torch.equal(m["colors"], want.colors_padded)
        if with_feats:
            assert torch.equal(m["features"], want.features_padded)
        else:
            assert m["features"] is None
        back = parallel.maps_to_pointclouds(m)
        for b in range(B):
            assert torch.equal(back.points_list[b], want.points_list[b])
            assert torch.equal(back.colors_list[b], want.colors_list[b])


def test_gather_maps_single_process():
    """No process group: the same function returns the local batch in the padded layout."""
    want = _whole_batch(3)
    m = parallel.gather_maps(want, 3)
    assert m["counts"] == [1, 4, 7]
    assert torch.equal(m["points"], want.points_padded) and torch.equal(m["features"], want.features_padded)
