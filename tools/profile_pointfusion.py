#!/usr/bin/env python3
"""Run a short PointFusion forward (BASELINE configs[2] shape) -- meant to be wrapped in rocprofv3."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
odom = sys.argv[2] if len(sys.argv) > 2 else "icp"
dev = "cuda:0"
c, d, K, P = make_sequence(1, n, 480, 640, seed=100)
frames = gs.RGBDImages(c.to(dev), d.to(dev), K.to(dev), P.to(dev))
slam = gs.slam.PointFusion(odom=odom, dsratio=4, numiters=10, device=dev)
import contextlib
from gradslam_amd.slam import icpslam as _icpslam
_t_enq = [0.0]
_orig_to_pc = _icpslam._MapArena.to_pointclouds
def _timed_to_pc(self):  # the frame loop has been enqueued when the map is asked for: host time up to here = enqueue time
    _t_enq[0] = time.perf_counter()
    return _orig_to_pc(self)
_icpslam._MapArena.to_pointclouds = _timed_to_pc
side = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("GS_SIDE_STREAM") else contextlib.nullcontext()
with torch.no_grad(), side:
    slam(gs.RGBDImages(c[:, :3].to(dev), d[:, :3].to(dev), K.to(dev), P[:, :3].to(dev)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pcs, poses = slam(frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("host enqueue ms/frame %.4f (the host has issued every launch of the sequence after this long)" % (1e3 * (_t_enq[0] - t0) / n))
print("frames/s", n / dt, "ms/frame", 1e3 * dt / n, "map", int(pcs.num_points_per_pointcloud.item()))
