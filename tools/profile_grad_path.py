#!/usr/bin/env python3
"""Where the differentiable PointFusion path spends its time (per frame phases + torch profiler table)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
odom = sys.argv[2] if len(sys.argv) > 2 else "gt"
dev = "cuda:0"
c, d, K, P = make_sequence(1, n, 480, 640, seed=7)


def run(profile):
    cc, dd, kk, pp = (x.to(dev).clone().requires_grad_(True) for x in (c, d, K, P))
    slam = gs.slam.PointFusion(odom=odom, dsratio=4, numiters=10, device=dev)
    frames = gs.RGBDImages(cc, dd, kk, pp)
    pcs = gs.Pointclouds(device=dev)
    prev = None
    for s in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        live = frames[:, s]
        poses = slam._localize(pcs, live, prev)
        live.poses = poses
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pcs = slam._map(pcs, live, True)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        prev = live if odom != "gt" else None
        if not profile:
            print("frame %d localize %.2f ms  map %.2f ms  N=%d" % (s, 1e3 * (t1 - t0), 1e3 * (t2 - t1), pcs.points_padded.shape[1]))
    loss = pcs.points_padded.sum() + pcs.colors_padded.mean()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    if not profile:
        print("backward %.2f ms total" % (1e3 * (time.perf_counter() - t0)))


run(False)
run(False)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(True)
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
