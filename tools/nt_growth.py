import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence
from gradslam_amd.slam.fusionutils import _project
dev = "cuda:0"
c, d, K, P = make_sequence(1, 40, 480, 640, seed=100)
frames = gs.RGBDImages(c.to(dev), d.to(dev), K.to(dev), P.to(dev))
slam = gs.slam.PointFusion(odom="icp", dsratio=4, numiters=10, device=dev)
pcs = gs.Pointclouds(device=dev); prev = None
with torch.no_grad():
    for s in range(40):
        live = frames[:, s]
        if prev is not None and s % 5 == 0:
            rows, cnt = _project(pcs, prev, 4)
            rows0, cnt0 = _project(pcs, prev, 0)
            print("frame", s, "map", pcs.points_padded.shape[1], "active", int(cnt0.item()), "ds-grid targets", int(cnt.item()))
        pcs, poses = slam.step(pcs, live, prev, inplace=True)
        live.poses = poses
        prev = live
