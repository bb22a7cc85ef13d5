#!/usr/bin/env python3
"""Measure (print, not assert) how close the HIP path gets to the reference goldens of config 1
(ref_slam_c1.npz, 64x64) and its less chaotic sibling (ref_slam_c1b.npz, 160x120): poses, map size, every
map attribute and all four input gradients.  The parity tests' tolerances are these numbers plus a margin."""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence

DEV = "cuda:0"
G = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def frac_off(a, b, tol):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double()
    return float(((a - b).abs() > tol * b.abs().max()).double().mean())


def inputs(name):
    g = dict(np.load(os.path.join(G, name + ".npz")))
    if "colors" in g:
        c = torch.from_numpy(g["colors"])
    else:
        L, H, W, seed = (int(x) for x in g["shape"])
        c = make_sequence(1, L, H, W, seed=seed)[0]
        assert float(c.double().sum()) == float(g["colors_sum"][0])
    return g, c, torch.from_numpy(g["depths"]), torch.from_numpy(g["intrinsics"]), torch.from_numpy(g["poses"])


for gname in ("ref_slam_c1", "ref_slam_c1b"):
    g, c, d, K, P = inputs(gname)
    for name, cls, odom in (("pf_gt", "PointFusion", "gt"), ("pf_icp", "PointFusion", "icp"), ("pf_gradicp", "PointFusion", "gradicp"),
                            ("is_gradicp", "ICPSLAM", "gradicp")):
        if name + "_poses" not in g:
            continue
        for rep in range(2):  # twice: the backward's float atomics are not bit-stable run to run
            cc, dd, kk, pp = (x.to(DEV).clone().requires_grad_(True) for x in (c, d, K, P))
            slam = getattr(gs.slam, cls)(odom=odom, dsratio=4, numiters=10, device=DEV)
            pcs, poses = slam(gs.RGBDImages(cc, dd, kk, pp))
            (poses.sum() + pcs.points_padded.sum() + pcs.colors_padded.mean()).backward()
            st = int(g[name + "_map_stride"][0]) if name + "_map_stride" in g else 1
            n_ref = int(g[name + "_map_count"][0]) if name + "_map_count" in g else g[name + "_map_points_0"].shape[0]
            n = pcs.points_list[0].shape[0]
            line = "%s %s rep%d pose %.2e map %d/%d" % (gname, name, rep, rel(poses.detach(), g[name + "_poses"]), n, n_ref)
            if n == n_ref:
                for attr, key in (("points_list", "points"), ("normals_list", "normals"), ("colors_list", "colors"), ("features_list", "feats")):
                    k = f"{name}_map_{key}_0"
                    if k in g and getattr(pcs, attr) is not None:
                        s = 1 if key == "feats" else st
                        line += " %s %.1e" % (key, rel(getattr(pcs, attr)[0].detach()[::s], g[k]))
            for k, x in (("colors", cc), ("depths", dd), ("intrinsics", kk), ("poses", pp)):
                got = x.grad if x.grad is not None else torch.zeros_like(x)
                ref = g[f"{name}_grad_{k}"]
                n_el = ref.size
                line += " | g_%s %.1e (elements off by >1e-3/1e-4/1e-5 of max: %d/%d/%d of %d)" % (
                    k, rel(got, ref), round(frac_off(got, ref, 1e-3) * n_el), round(frac_off(got, ref, 1e-4) * n_el),
                    round(frac_off(got, ref, 1e-5) * n_el), n_el)
            print(line, flush=True)
        with torch.no_grad():
            slam = getattr(gs.slam, cls)(odom=odom, dsratio=4, numiters=10, device=DEV)
            pcs, poses = slam(gs.RGBDImages(c.to(DEV), d.to(DEV), K.to(DEV), P.to(DEV)))
            print("   no-grad (streamed) pose %.2e map %d/%d" % (rel(poses, g[name + "_poses"]), pcs.points_list[0].shape[0], n_ref), flush=True)
