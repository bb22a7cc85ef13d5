#!/usr/bin/env python3
"""Streamed (arena) PointFusion forward vs the step-by-step path: equality + timing."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
odom = sys.argv[2] if len(sys.argv) > 2 else "icp"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = "cuda:0"
c, d, K, P = make_sequence(B, n, 480, 640, seed=100)
frames = gs.RGBDImages(c.to(dev), d.to(dev), K.to(dev), P.to(dev))
res = {}
for streamed in (True, False, True, False):
    slam = gs.slam.PointFusion(odom=odom, dsratio=4, numiters=10, device=dev)
    slam.streamed = streamed
    slam.fused_map = False  # the staged mapping step: the independent formulation
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pcs, poses = slam(frames)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("streamed" if streamed else "stepwise", "frames/s %.1f  ms/frame %.3f  map %s" % (n / dt, 1e3 * dt / n, pcs.num_points_per_pointcloud.tolist()))
    res[streamed] = (pcs, poses)
a, b = res[True], res[False]
print("poses equal", torch.equal(a[1], b[1]), "max diff", float((a[1] - b[1]).abs().max()))
for attr in ("points", "normals", "colors", "features"):
    for i in range(B):
        x, y = getattr(a[0], attr + "_list")[i], getattr(b[0], attr + "_list")[i]
        print(attr, i, x.shape == y.shape, bool(torch.equal(x, y)) if x.shape == y.shape else None)
