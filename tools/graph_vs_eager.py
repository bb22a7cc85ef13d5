#!/usr/bin/env python3
"""PointFusion forward at 640x480 with the ICP loops launched eagerly / replayed from the cached hipGraph: first frame
whose pose differs, per pair of runs (eager vs graph, graph vs graph, eager vs eager)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd import _native
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
c, d, K, P = make_sequence(1, n, 480, 640, seed=100)
frames = gs.RGBDImages(c.to(dev), d.to(dev), K.to(dev), P.to(dev))
runs = {}
for name, mode in (("eager", 0), ("graph", 1), ("graph2", 1), ("eager2", 0)):
    _native.lib().gs_set_graph_mode(mode)
    slam = gs.slam.PointFusion(odom="icp", dsratio=4, numiters=10, device=dev)
    with torch.no_grad():
        pcs, poses = slam(frames)
    torch.cuda.synchronize()
    runs[name] = (poses[0].clone(), int(pcs.num_points_per_pointcloud.item()))
    print(name, "map", runs[name][1], flush=True)
_native.lib().gs_set_graph_mode(-1)
for a, b in (("eager", "eager2"), ("graph", "graph2"), ("eager", "graph")):
    diff = (runs[a][0] != runs[b][0]).flatten(1).any(1).nonzero().flatten().tolist()
    mx = float((runs[a][0] - runs[b][0]).abs().max())
    print(a, "vs", b, ": first differing frames", diff[:6], "of", len(diff), "max abs pose diff %.3g" % mx)
