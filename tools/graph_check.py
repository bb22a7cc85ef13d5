#!/usr/bin/env python3
"""c2 localisation step: graph replay vs eager launches (timing + equality)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd import _native
from gradslam_amd.synthetic import make_sequence
dev = "cuda:0"
c, d, K, P = make_sequence(1, 4, 480, 640, seed=0)
slam = gs.slam.PointFusion(odom=sys.argv[1] if len(sys.argv) > 1 else "icp", dsratio=4, numiters=10, device=dev)
with torch.no_grad():
    f0 = gs.RGBDImages(c[:, :1].to(dev), d[:, :1].to(dev), K.to(dev), P[:, :1].to(dev))
    world, _ = slam.step(gs.Pointclouds(device=dev), f0, None)
    lives = [(c[:, s:s + 1].to(dev), d[:, s:s + 1].to(dev)) for s in (1, 2, 3)]
    Kd = K.to(dev)
    for mode in (1, 0, 1, 0):
        _native.lib().gs_set_graph_mode(mode)
        outs = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(60):
                cc, dd = lives[i % 3]
                pose = slam._localize(world, gs.RGBDImages(cc, dd, Kd), f0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
        print("graph" if mode else "eager", "ms/step %.4f" % (1e3 * dt), pose[0, 0, :3, 3].tolist())
