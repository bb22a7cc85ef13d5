#!/usr/bin/env python3
"""B sequences on ONE GPU against B x one sequence: PointFusion forward, 640x480 (a batch must not
pay for the padding of its arena).  usage: batch_scaling.py [frames=40] [B=4] [odom=icp]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
odom = sys.argv[3] if len(sys.argv) > 3 else "icp"
dev = "cuda:0"
c, d, K, P = (x.to(dev) for x in make_sequence(B, n, 480, 640, seed=100))
slam = gs.slam.PointFusion(odom=odom, dsratio=4, numiters=10, device=dev)


def run(sl):
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pcs, poses = slam(gs.RGBDImages(c[sl], d[sl], K[sl], P[sl]))
        torch.cuda.synchronize()
    return time.perf_counter() - t0, pcs, poses


run(slice(0, 1)); run(slice(0, B))  # warm the allocator
t1 = sum(run(slice(b, b + 1))[0] for b in range(B))
tb, pcs, poses = run(slice(0, B))
print("B = %d, %d frames, odom %s: batched %.1f ms (%.0f frames/s over all sequences), %d single runs %.1f ms (%.0f frames/s): batched / singles = %.3f"
      % (B, n, odom, 1e3 * tb, B * n / tb, B, 1e3 * t1, B * n / t1, tb / t1), "| maps", pcs.num_points_per_pointcloud.tolist())
