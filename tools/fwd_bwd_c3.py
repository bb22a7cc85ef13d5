#!/usr/bin/env python3
"""BASELINE configs[2] shape, reduced length: PointFusion forward + backward at 640x480 on the GPU."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
odom = sys.argv[2] if len(sys.argv) > 2 else "gradicp"
dev = "cuda:0"
c, d, K, P = make_sequence(1, n, 480, 640, seed=7)
for rep in range(3):
    pcs = poses = loss = None
    cc, dd, kk, pp = (x.to(dev).clone().requires_grad_(True) for x in (c, d, K, P))
    slam = gs.slam.PointFusion(odom=odom, dsratio=4, numiters=10, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pcs, poses = slam(gs.RGBDImages(cc, dd, kk, pp))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = poses.sum() + pcs.points_padded.sum() + pcs.colors_padded.mean()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rep", rep, "odom", odom, "frames", n, "fwd ms/frame %.2f" % (1e3 * (t1 - t0) / n), "bwd ms/frame %.2f" % (1e3 * (t2 - t1) / n),
          "map", int(pcs.num_points_per_pointcloud.item()),
          "grads finite", all(torch.isfinite(x.grad).all().item() for x in (cc, dd, kk, pp)),
          "|g_depth|", float(dd.grad.abs().sum()), "|g_pose|", float(pp.grad.abs().sum()))
