#!/usr/bin/env python3
"""Generate tests/golden/ref_slam_fixture.npz by importing the UNMODIFIED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_fixture.py

Full SLAM on the reference's own REAL-SENSOR fixture (tests/data/msrd_b2s3: B = 2 sequences of L = 3 frames,
160x120, 11.8 % depth holes, fy < 0 -- the inputs are already in tests/golden/msrd_b2s3.npz) through
gradslam/slam/icpslam.py:99-138 and slam/pointfusion.py:107-112:  PointFusion and ICPSLAM x {icp, gradicp},
dsratio 4, 10 iterations.  Written per case: the recovered poses, per-sequence map sizes, a strided sample of every
map attribute with fp64 checksums of the whole arrays, and the reference's own autograd gradients of
poses.sum() + points_padded.sum() + colors_padded.mean()  with respect to colours (strided), depths, intrinsics and
poses.  Stand-ins as in tools/gen_golden.py (tools/oracle_shims/README.md).
"""
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path[:0] = [os.path.join(REPO, "tools", "oracle_shims"), REF]

import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
torch.manual_seed(0)

import gradslam  # noqa: E402,F401  (the reference)
from gradslam.slam.icpslam import ICPSLAM  # noqa: E402
from gradslam.slam.pointfusion import PointFusion  # noqa: E402
from gradslam.structures.rgbdimages import RGBDImages  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
npy = lambda t: t.detach().cpu().numpy()
fx = {k: np.load(os.path.join(REF, "tests", "data", "msrd_b2s3", k + ".npy")) for k in
      ("colors", "depths", "intrinsics", "poses")}
CASES = [("pf_icp", PointFusion, "icp", 3), ("pf_gradicp", PointFusion, "gradicp", 3),
         ("is_icp", ICPSLAM, "icp", 7), ("is_gradicp", ICPSLAM, "gradicp", 7)]
CSTRIDE = 5
S = {"color_grad_stride": np.array([CSTRIDE])}
for name, cls, odom, st in CASES:
    cc, dd, kk, pp = (torch.from_numpy(fx[k]).float().clone().requires_grad_(True)
                      for k in ("colors", "depths", "intrinsics", "poses"))
    slam = cls(odom=odom, dsratio=4, numiters=10)
    pcs, poses = slam(RGBDImages(cc, dd, kk, pp))
    (poses.sum() + pcs.points_padded.sum() + pcs.colors_padded.mean()).backward()
    S[name + "_poses"] = npy(poses)
    S[name + "_counts"] = npy(pcs.num_points_per_pointcloud).astype(np.int64)
    S[name + "_map_stride"] = np.array([st])
    lists = [("points", pcs.points_list), ("normals", pcs.normals_list), ("colors", pcs.colors_list)]
    if pcs.has_features:
        lists.append(("feats", pcs.features_list))
    for b in range(len(pcs)):
        for attr, lst in lists:
            a = npy(lst[b])
            S[f"{name}_map_{attr}_{b}"] = a[::st]
            S[f"{name}_map_{attr}_{b}_sum"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
    g0 = lambda t: npy(t.grad if t.grad is not None else torch.zeros_like(t))
    S[name + "_grad_colors"] = g0(cc).reshape(-1, 3)[::CSTRIDE]
    S[name + "_grad_colors_sum"] = np.array([g0(cc).astype(np.float64).sum(), np.abs(g0(cc).astype(np.float64)).sum()])
    S[name + "_grad_depths"] = g0(dd)
    S[name + "_grad_intrinsics"] = g0(kk)
    S[name + "_grad_poses"] = g0(pp)
    print(name, "maps", S[name + "_counts"].tolist(), "pose move vs input",
          float((poses.detach() - pp.detach()).abs().max()), flush=True)
np.savez_compressed(os.path.join(OUT, "ref_slam_fixture.npz"), **S)
print("ref_slam_fixture.npz", os.path.getsize(os.path.join(OUT, "ref_slam_fixture.npz")) // 1024, "KiB")
