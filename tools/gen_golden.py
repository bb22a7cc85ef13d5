#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the UNMODIFIED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

The reference (/root/reference) never travels to the GPU box; these small fixtures do.  The
third-party modules it imports that are absent here are replaced by the stand-ins documented in
tools/oracle_shims/README.md (nearest-neighbour search = our own brute force: indices are
parity-unpinned; 4x4 rigid algebra corroborated by the reference's own equivalents).

Fixtures written
  msrd_b2s3.npz      the reference's own test fixture tests/data/msrd_b2s3/*.npy (inputs AND the
                     golden vertex/normal/global maps its tests compare against) -- data, copied.
  ref_units.npz      per-function outputs of the reference on that fixture (projection tables,
                     similar masks, unique correspondences, fused map, alpha, se3_exp, downsample).
  ref_icp_trace.npz  per-iteration trace (AtA, Atb, err, new_err, damp, xi) + final T for ICP and
                     gradICP on a 64x64 synthetic pair and the fixture pair of test_icp.py.
  ref_slam_c1.npz    BASELINE config 1: 2-frame 64x64 synthetic, B=1, PointFusion x {gt,icp,gradicp}
                     and ICPSLAM x {gradicp}: poses, final map attributes, and input gradients of
                     poses.sum() + points.sum() + colors.mean().
"""
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path[:0] = [os.path.join(REPO, "tools", "oracle_shims"), REF]

import importlib.util
import math
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
torch.manual_seed(0)

import gradslam  # noqa: E402  (the reference)
from gradslam.odometry import icputils as R_icp  # noqa: E402
from gradslam.slam import fusionutils as R_fus  # noqa: E402
from gradslam.slam.icpslam import ICPSLAM  # noqa: E402
from gradslam.slam.pointfusion import PointFusion  # noqa: E402
from gradslam.structures.pointclouds import Pointclouds  # noqa: E402
from gradslam.structures.rgbdimages import RGBDImages  # noqa: E402
from gradslam.structures.utils import pointclouds_from_rgbdimages  # noqa: E402
from gradslam.geometry.se3utils import se3_exp  # noqa: E402

_spec = importlib.util.spec_from_file_location("syn", os.path.join(REPO, "gradslam_amd", "synthetic.py"))
syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(syn)

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
npy = lambda t: t.detach().cpu().numpy()


def fixture():
    d = os.path.join(REF, "tests", "data", "msrd_b2s3")
    return {k: np.load(os.path.join(d, k + ".npy")) for k in
            ["colors", "depths", "intrinsics", "poses", "vertex_map", "normal_map",
             "global_vertex_map", "global_normal_map"]}


def cloud_dict(prefix, pc):
    out = {}
    for b in range(len(pc)):
        out[f"{prefix}_points_{b}"] = npy(pc.points_list[b])
        out[f"{prefix}_normals_{b}"] = npy(pc.normals_list[b])
        out[f"{prefix}_colors_{b}"] = npy(pc.colors_list[b])
        if pc.has_features:
            out[f"{prefix}_feats_{b}"] = npy(pc.features_list[b])
    return out


# ------------------------------------------------------------------ fixture copy
fx = fixture()
np.savez_compressed(os.path.join(OUT, "msrd_b2s3.npz"), **fx)

# ------------------------------------------------------------------ unit vectors
U = {}
colors, depths, K, poses = (torch.from_numpy(fx[k]) for k in ["colors", "depths", "intrinsics", "poses"])
rgbd = RGBDImages(colors, depths, K, poses)
f0, f1 = rgbd[:, 0], rgbd[:, 1]
# map = PointFusion map after frame 0 (all valid pixels, ccount = alpha)
sigma, dist_th, dot_th = 0.6, 0.05, math.cos(math.radians(20))
pc = R_fus.update_map_fusion(Pointclouds(), f0, dist_th, dot_th, sigma)
U.update(cloud_dict("map0", pc))
t_act = R_fus.find_active_map_points(pc, f1)
t_sim, m_sim = R_fus.find_similar_map_points(pc, f1, t_act, dist_th, dot_th)
t_uni = R_fus.find_best_unique_correspondences(pc, f1, t_sim)
U["active_f1"], U["similar_f1"], U["similar_mask_f1"], U["unique_f1"] = map(npy, (t_act, t_sim, m_sim, t_uni))
pc1 = R_fus.fuse_with_map(pc.clone(), f1, t_uni, sigma)
U.update(cloud_dict("map1", pc1))
pc2 = R_fus.update_map_fusion(pc1.clone(), rgbd[:, 2], dist_th, dot_th, sigma)
U["map2_counts"] = npy(pc2.num_points_per_pointcloud)
U["map2_sums"] = np.stack([np.stack([npy(x[b].double().sum(0))[:1].repeat(3) if x[b].shape[1] == 1 else npy(x[b].double().sum(0))
                                      for x in (pc2.points_list, pc2.normals_list, pc2.colors_list, pc2.features_list)])
                           for b in range(len(pc2))])
U["alpha_f1"] = npy(R_fus.get_alpha(f1.vertex_map, dim=4, keepdim=True, sigma=sigma))
# downsampling (ds=4) of the live frame and of the active map
fr_ds = R_icp.downsample_rgbdimages(f1, 4)
mp_ds = R_icp.downsample_pointclouds(pc, R_fus.find_active_map_points(pc, f0), 4)
U.update(cloud_dict("frame_ds4", fr_ds))
U.update(cloud_dict("mapds4", mp_ds))
# aggregate map (ICPSLAM mapping)
agg = R_fus.update_map_aggregate(R_fus.update_map_aggregate(Pointclouds(), f0), f1)
U["agg01_counts"] = npy(agg.num_points_per_pointcloud)
U["agg01_point_sums"] = np.stack([npy(agg.points_list[b].double().sum(0)) for b in range(len(agg))])
# se3_exp: small-angle, general, large
xis = torch.tensor([[0.01, -0.02, 0.03, 1e-8, -2e-8, 3e-8],
                    [0.05, 0.03, 0.01, 0.1, -0.2, 0.05],
                    [1.0, -2.0, 0.5, 1.2, 0.7, -2.1],
                    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]])
U["se3_xi"] = npy(xis)
U["se3_T"] = np.stack([npy(se3_exp(x.view(6, 1))) for x in xis])
np.savez_compressed(os.path.join(OUT, "ref_units.npz"), **U)

# ------------------------------------------------------------------ ICP traces
TR = {}


def traced(fn, *args, **kw):
    """Run a reference ICP routine while recording what its own helpers computed."""
    rec = []
    g0, s0 = R_icp.gauss_newton_solve, R_icp.solve_linear_system
    calls = {"gn": []}

    def gn(*a, **k):
        A, b, idx = g0(*a, **k)
        calls["gn"].append((A.detach(), b.detach(), idx.detach()))
        return A, b, idx

    def sl(A, b, damp):
        x = s0(A, b, damp)
        rec.append(dict(AtA=npy(A.t() @ A), Atb=npy(A.t() @ b)[:, 0], damp=float(damp), xi=npy(x)[:, 0],
                        err=float((b[:, 0] * b[:, 0]).sum()), n=A.shape[0], idx=npy(calls["gn"][-1][2])))
        return x

    R_icp.gauss_newton_solve, R_icp.solve_linear_system = gn, sl
    try:
        T, idx = fn(*args, **kw)
    finally:
        R_icp.gauss_newton_solve, R_icp.solve_linear_system = g0, s0
    # every iteration = [first solve, look-ahead]; new_err from the look-ahead's b
    for i, r in enumerate(rec):
        b1 = calls["gn"][2 * i + 1][1]
        r["new_err"] = float((b1[:, 0] * b1[:, 0]).sum())
    return T, idx, rec


def pack_trace(prefix, T, idx, rec):
    TR[prefix + "_T"] = npy(T)
    TR[prefix + "_idx_last"] = npy(idx)
    for k in ["AtA", "Atb", "xi"]:
        TR[f"{prefix}_{k}"] = np.stack([r[k] for r in rec])
    for k in ["damp", "err", "new_err", "n"]:
        TR[f"{prefix}_{k}"] = np.array([r[k] for r in rec], dtype=np.float64)
    TR[prefix + "_idx0"] = rec[0]["idx"]


# (a) 64x64 synthetic pair, ds=1 clouds: src = frame 1 (posed with frame 0's pose), tgt = frame 0
c, d, Ks, Ps = syn.make_sequence(1, 2, 64, 64, seed=0)
r = RGBDImages(c, d, Ks, Ps[:, :1].repeat(1, 2, 1, 1))
tgt = pointclouds_from_rgbdimages(r[:, 0])
src = pointclouds_from_rgbdimages(r[:, 1])
TR["syn_src"], TR["syn_tgt"], TR["syn_tgt_n"] = npy(src.points_list[0]), npy(tgt.points_list[0]), npy(tgt.normals_list[0])
args = (src.points_list[0][None], tgt.points_list[0][None], tgt.normals_list[0][None], torch.eye(4))
pack_trace("syn_icp", *traced(R_icp.point_to_plane_ICP, *args, numiters=10, damp=1e-8, dist_thresh=None))
pack_trace("syn_icp_th", *traced(R_icp.point_to_plane_ICP, *args, numiters=10, damp=1e-8, dist_thresh=0.01))
pack_trace("syn_gradicp", *traced(R_icp.point_to_plane_gradICP, *args, numiters=10, damp=1e-8, dist_thresh=None))

# (b) the fixture case of reference tests/odometry/test_icp.py:14-53 (rad=0.1, 30 iters, thresh 0.2)
r1 = RGBDImages(colors[:1], depths[:1], K[:1], poses[:1])
srcp = pointclouds_from_rgbdimages(r1[:, 0])
rad = 0.1
Tgt = torch.tensor([[np.cos(rad), -np.sin(rad), 0.0, 0.05], [np.sin(rad), np.cos(rad), 0.0, 0.03],
                    [0.0, 0.0, 1.0, 0.01], [0.0, 0.0, 0.0, 1.0]], dtype=colors.dtype)
tgtp = srcp.transform(Tgt)
TR["fix_T_true"] = npy(Tgt)
TR["fix_src"], TR["fix_tgt"], TR["fix_tgt_n"] = npy(srcp.points_list[0]), npy(tgtp.points_list[0]), npy(tgtp.normals_list[0])
args = (srcp.points_list[0][None], tgtp.points_list[0][None], tgtp.normals_list[0][None], torch.eye(4))
pack_trace("fix_icp", *traced(R_icp.point_to_plane_ICP, *args, numiters=30, damp=1e-8, dist_thresh=0.2))
pack_trace("fix_gradicp", *traced(R_icp.point_to_plane_gradICP, *args, numiters=30, damp=1e-8, dist_thresh=0.2))
np.savez_compressed(os.path.join(OUT, "ref_icp_trace.npz"), **TR)

# ------------------------------------------------------------------ config 1 + gradients
S = {}
c, d, Ks, Ps = syn.make_sequence(1, 2, 64, 64, seed=0)
S["colors"], S["depths"], S["intrinsics"], S["poses"] = map(npy, (c, d, Ks, Ps))
for name, cls, odom in [("pf_gt", PointFusion, "gt"), ("pf_icp", PointFusion, "icp"),
                        ("pf_gradicp", PointFusion, "gradicp"), ("is_gradicp", ICPSLAM, "gradicp")]:
    cc, dd, kk, pp = (x.clone().requires_grad_(True) for x in (c, d, Ks, Ps))
    slam = cls(odom=odom, dsratio=4, numiters=10)
    pcs, rec_poses = slam(RGBDImages(cc, dd, kk, pp))
    S[name + "_poses"] = npy(rec_poses)
    S.update(cloud_dict(name + "_map", pcs))
    loss = rec_poses.sum() + pcs.points_padded.sum() + pcs.colors_padded.mean()
    loss.backward()
    for gname, t in [("colors", cc), ("depths", dd), ("intrinsics", kk), ("poses", pp)]:
        S[f"{name}_grad_{gname}"] = npy(t.grad if t.grad is not None else torch.zeros_like(t))
np.savez_compressed(os.path.join(OUT, "ref_slam_c1.npz"), **S)

for f in sorted(os.listdir(OUT)):
    print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
