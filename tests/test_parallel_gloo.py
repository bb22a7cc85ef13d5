"""World-size-2 gloo test of the one-sequence-per-GPU sharding (CPU processes stand in for ranks)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from gradslam_amd import parallel


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


class _FakeMap:
    def __init__(self, pts):
        self.points_list = pts


def _worker(rank, world, port, B, L, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    colors = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1, 1).expand(B, L, 2, 2, 3).contiguous()

    def slam_fn(c, d, k, p):  # per-sequence "SLAM": pose (b, l) encodes (sequence id, frame id)
        ids = c[:, 0, 0, 0, 0]
        poses = torch.eye(4).view(1, 1, 4, 4).repeat(len(ids), L, 1, 1)
        poses[:, :, 0, 3] = ids.view(-1, 1)
        poses[:, :, 1, 3] = torch.arange(L, dtype=torch.float32).view(1, -1)
        return _FakeMap([torch.full((int(i) + 1, 3), float(i)) for i in ids]), poses

    _, all_poses, maps = parallel.run_sharded(slam_fn, colors, None, None, None, gather_maps=True)
    torch.save({"poses": all_poses, "maps": maps}, os.path.join(out_dir, "r{}.pt".format(rank)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("B", [2, 5])
def test_sharded_gather_world2(tmp_path, B):
    L, world = 3, 2
    mp.spawn(_worker, args=(world, _free_port(), B, L, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "r{}.pt".format(r))) for r in range(world)]
    for o in outs:  # every rank holds the whole batch, in batch order
        assert o["poses"].shape == (B, L, 4, 4)
        assert o["poses"][:, 0, 0, 3].tolist() == [float(b) for b in range(B)]
        assert o["poses"][0, :, 1, 3].tolist() == [0.0, 1.0, 2.0]
    # ragged map gather: rank r contributed the concatenation of its sequences' points
    sizes = [m.shape[0] for m in outs[0]["maps"]]
    want = [sum(i + 1 for i in parallel.shard_indices(B, world, r)) for r in range(world)]
    assert sizes == want
