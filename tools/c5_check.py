#!/usr/bin/env python3
"""BASELINE configs[4] shape: ICPSLAM (gradicp) on 1296x968 frames -- forward frames/s and fwd+bwd per frame."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence_cached as make_sequence
dev = "cuda:0"
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
c, d, K, P = make_sequence(1, L, 968, 1296, seed=11)
frames = gs.RGBDImages(c.to(dev), d.to(dev), K.to(dev), P.to(dev))
for rep in range(3):
    slam = gs.slam.ICPSLAM(odom="gradicp", dsratio=4, numiters=10, device=dev)
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pcs, poses = slam(frames)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("forward: %.2f ms/frame (%.0f frames/s), map %d, pose err %.4f" % (1e3 * dt / L, L / dt, int(pcs.num_points_per_pointcloud.item()), float((poses.cpu() - P).abs().max())))
for rep in range(2):
    leaves = [x.to(dev).clone().requires_grad_(True) for x in (d, K, P)]
    slam = gs.slam.ICPSLAM(odom="gradicp", dsratio=4, numiters=10, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pcs, poses = slam(gs.RGBDImages(c.to(dev), *leaves))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (poses[:, -1, :3, 3].sum() + pcs.points_padded.mean()).backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("with gradients: fwd %.2f + bwd %.2f ms/frame, grads finite %s" % (1e3 * (t1 - t0) / L, 1e3 * (t2 - t1) / L, all(bool(torch.isfinite(x.grad).all()) for x in leaves)))
