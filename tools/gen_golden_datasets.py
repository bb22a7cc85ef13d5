#!/usr/bin/env python3
"""Generate tests/golden/ref_datasets.npz from the UNMODIFIED reference's dataset helpers (build container
only): datasets/datautils.py and datasets/tumutils.py are loaded straight from their files (the package's
datasets/__init__ imports cv2 / imageio, absent here; these two modules need numpy + torch only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_datasets.py
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


R_du = load("ref_datautils", "/root/reference/gradslam/datasets/datautils.py")
R_tu = load("ref_tumutils", "/root/reference/gradslam/datasets/tumutils.py")
R_tu.sys = sys  # the module writes to sys.stderr without importing sys
rng = np.random.default_rng(7)
G = {}
G["pq"] = rng.normal(size=(40, 7))
G["pq_T"] = R_du.pointquaternion_to_homogeneous(G["pq"].copy())
G["K"] = np.array([[525.0, 0, 319.5, 0], [0, 525.0, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
G["K_scaled"] = R_du.scale_intrinsics(G["K"], 0.25, 0.5)
G["poses"] = np.stack([R_du.pointquaternion_to_homogeneous(p).astype(np.float64) for p in rng.normal(size=(6, 7))])
G["poses_transforms"] = np.stack(R_du.poses_to_transforms(list(G["poses"])))
# time-stamp association: stamps of two unsynchronised streams + a trajectory file
G["stamps_a"] = np.sort(rng.uniform(0, 20, 200)).round(6)
G["stamps_b"] = np.sort(rng.uniform(0, 20, 230)).round(6)
G["traj"] = np.concatenate([np.sort(rng.uniform(0, 20, 400)).round(4)[:, None], rng.normal(size=(400, 7))], 1)
d = tempfile.mkdtemp()
fa, fb, ft = (os.path.join(d, n) for n in ("rgb.txt", "depth.txt", "groundtruth.txt"))
open(fa, "w").write("# colour\n" + "\n".join("%.6f rgb/%.6f.png" % (t, t) for t in G["stamps_a"]) + "\n")
open(fb, "w").write("# depth\n" + "\n".join("%.6f depth/%.6f.png" % (t, t) for t in G["stamps_b"]) + "\n")
open(ft, "w").write("# gt\n" + "\n".join("%.4f %.6f %.6f %.6f %.6f %.6f %.6f %.6f" % tuple(r) for r in G["traj"]) + "\n")
da, db = R_tu.read_file_list(fa, 3, 150), R_tu.read_file_list(fb)
for tag, off, md in (("m0", 0.0, 0.02), ("m1", 0.013, 0.05), ("m2", -0.02, 0.3)):
    m = R_tu.associate(da, db, off, md)
    G[tag + "_a"] = np.array([float(x) for x, _ in m])
    G[tag + "_b"] = np.array([float(y) for _, y in m])
tr = R_tu.read_trajectory(ft, matrix=True)
G["traj_keys"] = np.array([float(k) for k in tr.keys()])
G["traj_T"] = np.stack(list(tr.values()))
out = os.path.join(REPO, "tests", "golden", "ref_datasets.npz")
np.savez_compressed(out, **G)
print("wrote", out, os.path.getsize(out) // 1024, "KiB")
