#!/usr/bin/env python3
"""Sanity of the RCCL calls bench.py / gradslam_amd.parallel make when WORLD_SIZE > 1, on a one-rank "nccl" group
(the only RCCL configuration a one-GPU box can run): init with device_id, all_gather of pose blocks in HBM,
MAX all_reduce of a float64 scalar, barrier."""
import os, sys
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
pad = torch.eye(4, device=dev).repeat(1, 5, 1, 1)
out = [torch.empty_like(pad)]
dist.all_gather(out, pad)
assert torch.equal(out[0], pad)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
n = torch.tensor([7], dtype=torch.int64, device=dev)
sizes = [torch.zeros_like(n)]
dist.all_gather(sizes, n)
assert int(sizes[0].item()) == 7
dist.barrier()
torch.cuda.synchronize()
print("rccl one-rank group: backend", dist.get_backend(), "all_gather / all_reduce(MAX, f64) / barrier ok")
dist.destroy_process_group()
