#!/usr/bin/env python3
"""Generate tests/golden/ref_slam_c1b.npz by importing the UNMODIFIED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_c1b.py [--sensitivity]

BASELINE config 1's plumbing at a size where the LM loop is no longer chaotic: 3-frame 160x120 synthetic
RGB-D, B=1, dsratio 4 (1 200 ICP points instead of config 1's 256), 10 iterations.  For PointFusion x
{icp, gradicp} and ICPSLAM x {gradicp}: recovered poses, map size, every map attribute, and the reference's
own autograd gradients of  poses.sum() + points.sum() + colors.mean()  with respect to colours, depths,
intrinsics and poses.  Same stand-ins as tools/gen_golden.py (tools/oracle_shims/README.md).

--sensitivity additionally prints how far the REFERENCE's own outputs move when the depth is perturbed by
1e-7 relative (what a parity tolerance at this size can mean); nothing is written in that mode.
"""
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path[:0] = [os.path.join(REPO, "tools", "oracle_shims"), REF]

import importlib.util
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
torch.manual_seed(0)

import gradslam  # noqa: E402,F401  (the reference)
from gradslam.slam.icpslam import ICPSLAM  # noqa: E402
from gradslam.slam.pointfusion import PointFusion  # noqa: E402
from gradslam.structures.rgbdimages import RGBDImages  # noqa: E402

_spec = importlib.util.spec_from_file_location("syn", os.path.join(REPO, "gradslam_amd", "synthetic.py"))
syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(syn)

OUT = os.path.join(REPO, "tests", "golden")
npy = lambda t: t.detach().cpu().numpy()
CASES = [("pf_icp", PointFusion, "icp"), ("pf_gradicp", PointFusion, "gradicp"), ("is_gradicp", ICPSLAM, "gradicp")]
H, W, L, SEED = 120, 160, 3, 11


def run(cls, odom, c, d, K, P):
    cc, dd, kk, pp = (x.clone().requires_grad_(True) for x in (c, d, K, P))
    slam = cls(odom=odom, dsratio=4, numiters=10)
    pcs, poses = slam(RGBDImages(cc, dd, kk, pp))
    (poses.sum() + pcs.points_padded.sum() + pcs.colors_padded.mean()).backward()
    grads = {g: npy(t.grad if t.grad is not None else torch.zeros_like(t))
             for g, t in (("colors", cc), ("depths", dd), ("intrinsics", kk), ("poses", pp))}
    return pcs, poses, grads


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


c, d, K, P = syn.make_sequence(1, L, H, W, seed=SEED)
if "--sensitivity" in sys.argv:
    for name, cls, odom in CASES:
        p0, q0, g0 = run(cls, odom, c, d, K, P)
        d2 = d * (1.0 + 1e-7 * torch.sign(torch.randn(d.shape, generator=torch.Generator().manual_seed(1))))
        p1, q1, g1 = run(cls, odom, c, d2, K, P)
        print(name, "pose moves by", rel(npy(q1), npy(q0)), "map size", p0.points_list[0].shape[0], "->", p1.points_list[0].shape[0],
              {k: rel(g1[k], g0[k]) for k in g0})
    sys.exit(0)

# inputs: the colours are uniform noise (incompressible); tests regenerate them with the same generator
# (gradslam_amd.synthetic.make_sequence(1, L, H, W, seed=SEED)) and check this checksum
S = {"depths": npy(d), "intrinsics": npy(K), "poses": npy(P), "colors_sum": np.array([float(c.double().sum())]),
     "shape": np.array([L, H, W, SEED])}
for name, cls, odom in CASES:
    pcs, poses, grads = run(cls, odom, c, d, K, P)
    S[name + "_poses"] = npy(poses)
    # the aggregate map of ICPSLAM is every valid pixel of every frame: a strided sample pins it
    st = 1 if pcs.has_features else 4
    S[name + "_map_count"] = np.array([pcs.points_list[0].shape[0]])
    S[name + "_map_stride"] = np.array([st])
    S[name + "_map_points_0"] = npy(pcs.points_list[0])[::st]
    S[name + "_map_normals_0"] = npy(pcs.normals_list[0])[::st]
    S[name + "_map_colors_0"] = npy(pcs.colors_list[0])[::st]
    if pcs.has_features:
        S[name + "_map_feats_0"] = npy(pcs.features_list[0])
    for g, v in grads.items():
        S[f"{name}_grad_{g}"] = v
    print(name, "map", pcs.points_list[0].shape[0], "pose err vs gt", rel(npy(poses), npy(P)))
np.savez_compressed(os.path.join(OUT, "ref_slam_c1b.npz"), **S)
print("ref_slam_c1b.npz", os.path.getsize(os.path.join(OUT, "ref_slam_c1b.npz")) // 1024, "KiB")
