#!/usr/bin/env python3
"""Generate tests/golden/ref_icp_grads.npz by importing the UNMODIFIED reference (build container only):
input gradients of the reference's point_to_plane_ICP / point_to_plane_gradICP as torch autograd derives
them -- the known answers for the fused reverse pass (gs_icp_point_to_plane_backward).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_icp_grads.py

Cases: the 64x64 synthetic pair of ref_icp_trace.npz (src = frame 1 posed with frame 0's pose, tgt =
frame 0, ds=1 clouds), a non-trivial initial transform, loss = sum(T * W) for a fixed W, gradients with
respect to src, tgt, tgt normals and the initial transform; few iterations (the loss is a smooth function
of the inputs only while the association and the accept decisions stay put).  Same stand-ins as
tools/gen_golden.py (tools/oracle_shims/README.md).
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
from gradslam.odometry import icputils as R_icp  # noqa: E402
from gradslam.structures.rgbdimages import RGBDImages  # noqa: E402
from gradslam.structures.utils import pointclouds_from_rgbdimages  # noqa: E402

_spec = importlib.util.spec_from_file_location("syn", os.path.join(REPO, "gradslam_amd", "synthetic.py"))
syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(syn)

OUT = os.path.join(REPO, "tests", "golden")
npy = lambda t: t.detach().cpu().numpy()

c, d, Ks, Ps = syn.make_sequence(1, 2, 64, 64, seed=0)
r = RGBDImages(c, d, Ks, Ps[:, :1].repeat(1, 2, 1, 1))
tgt = pointclouds_from_rgbdimages(r[:, 0])
src = pointclouds_from_rgbdimages(r[:, 1])
G = {"src": npy(src.points_list[0]), "tgt": npy(tgt.points_list[0]), "tgt_n": npy(tgt.normals_list[0])}
W = torch.tensor([[0.3, -1.1, 0.7, 2.0], [1.3, 0.4, -0.6, -1.5], [-0.8, 0.9, 0.2, 1.0], [0.0, 0.0, 0.0, 0.0]])
rad = 0.004
T0 = torch.tensor([[np.cos(rad), -np.sin(rad), 0.0, 0.002], [np.sin(rad), np.cos(rad), 0.0, -0.001], [0.0, 0.0, 1.0, 0.003],
                   [0.0, 0.0, 0.0, 1.0]], dtype=torch.float32)
G["W"], G["T0"] = npy(W), npy(T0)

CASES = [("icp_n1", R_icp.point_to_plane_ICP, dict(numiters=1, damp=1e-8, dist_thresh=None)),
         ("icp_n4", R_icp.point_to_plane_ICP, dict(numiters=4, damp=1e-8, dist_thresh=None)),
         ("icp_n4_th", R_icp.point_to_plane_ICP, dict(numiters=4, damp=1e-8, dist_thresh=2e-4)),
         ("icp_n3_damp", R_icp.point_to_plane_ICP, dict(numiters=3, damp=1e-2, dist_thresh=None)),
         ("gradicp_n1", R_icp.point_to_plane_gradICP, dict(numiters=1, damp=1e-8, dist_thresh=None)),
         ("gradicp_n3", R_icp.point_to_plane_gradICP, dict(numiters=3, damp=1e-8, dist_thresh=None)),
         ("gradicp_n3_th", R_icp.point_to_plane_gradICP, dict(numiters=3, damp=1e-8, dist_thresh=2e-4)),
         ("gradicp_n3_damp", R_icp.point_to_plane_gradICP, dict(numiters=3, damp=1e-2, dist_thresh=None, lambda_max=3.0, B=0.7,
                                                                  B2=1.3, nu=50.0))]
for name, fn, kw in CASES:
    s, t, n, T = (torch.tensor(G[k]).clone().requires_grad_(True) for k in ("src", "tgt", "tgt_n", "T0"))
    Tout, idx = fn(s[None], t[None], n[None], T, **kw)
    (Tout * W).sum().backward()
    G[name + "_T"] = npy(Tout)
    for k, v in (("g_src", s), ("g_tgt", t), ("g_nrm", n), ("g_T0", T)):
        G[name + "_" + k] = npy(v.grad)
    print(name, "T err vs eye %.3e" % float((Tout - torch.eye(4)).abs().max()),
          "|g_src| %.3e |g_tgt| %.3e |g_nrm| %.3e |g_T0| %.3e" % tuple(float(v.grad.abs().sum()) for v in (s, t, n, T)))
np.savez_compressed(os.path.join(OUT, "ref_icp_grads.npz"), **G)
print("wrote", os.path.join(OUT, "ref_icp_grads.npz"), os.path.getsize(os.path.join(OUT, "ref_icp_grads.npz")) // 1024, "KiB")
