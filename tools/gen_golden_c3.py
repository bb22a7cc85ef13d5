#!/usr/bin/env python3
"""Generate tests/golden/ref_slam_c3.npz by importing the UNMODIFIED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 GS_SHIM_KNN=c python tools/gen_golden_c3.py [L]

BASELINE configs[2]'s shape at a length the reference can be run at on the CPU: PointFusion, synthetic TUM-shape
640x480, B = 1, dsratio 4, 10 iterations, odom in {icp, gradicp}, L = 64 frames -- long enough for the ICP target
(the downsampled active map) to pass four points per ds-grid pixel and for the map to pass 4 H W points, i.e. the
regime in which the HIP path switches to its grid search and small tiles.  Written per case: every recovered pose,
the map size after every frame, a strided sample of the final map's attributes and fp64 checksums of all of them, and
the reference's OWN sensitivity (pose and map-size deviation per frame under a 1e-7 relative depth perturbation).
The inputs are NOT stored (gradslam_amd.synthetic.make_sequence(1, L, 480, 640, seed=SEED) regenerates them; a
checksum pins that).  Stand-ins as in tools/gen_golden.py; the nearest-neighbour stand-in runs oracle/knn_ref.c
(GS_SHIM_KNN=c: our own code either way, see tools/oracle_shims/README.md).
"""
import os
import sys
import time

sys.dont_write_bytecode = True
os.environ.setdefault("GS_SHIM_KNN", "c")
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
from gradslam.slam.pointfusion import PointFusion  # noqa: E402
from gradslam.structures.rgbdimages import RGBDImages  # noqa: E402

_spec = importlib.util.spec_from_file_location("syn", os.path.join(REPO, "gradslam_amd", "synthetic.py"))
syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(syn)

OUT = os.path.join(REPO, "tests", "golden")
npy = lambda t: t.detach().cpu().numpy()
H, W, SEED, STRIDE = 480, 640, 5, 97
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64

c, d, K, P = syn.make_sequence(1, L, H, W, seed=SEED)
S = {"shape": np.array([L, H, W, SEED]), "stride": np.array([STRIDE]),
     "depths_sum": np.array([float(d.double().sum())]), "colors_sum": np.array([float(c.double().sum())]),
     "intrinsics": npy(K), "poses_gt": npy(P)}
def run_reference(odom, depth):
    slam = PointFusion(odom=odom, dsratio=4, numiters=10)
    counts, t0 = [], time.time()
    inner = slam._map                      # the reference's own bound method; the wrapper only records the map size

    def recording_map(pointclouds, live_frame, inplace=False, _inner=inner):
        out = _inner(pointclouds, live_frame, inplace)
        counts.append(int(out.num_points_per_pointcloud[0]))
        print(odom, "frame", len(counts), "map", counts[-1], "%.0f s" % (time.time() - t0), flush=True)
        return out

    slam._map = recording_map
    with torch.no_grad():
        pcs, poses = slam(RGBDImages(c, depth, K, P))
    return pcs, poses, counts


# What a parity bound over 64 frames can mean: the REFERENCE's own response to a 1e-7 relative perturbation of the depth
# (every pixel multiplied by 1 +- 1e-7, seeded signs) -- per-frame pose deviation (max abs over the 4x4, relative to the
# frame's largest pose entry) and map-size deviation.  Ten LM iterations from the identity on a weakly constrained surface
# amplify a last-bit difference frame after frame; the tests derive their whole-sequence bounds from these arrays.
d_pert = d * (1.0 + 1e-7 * torch.sign(torch.randn(d.shape, generator=torch.Generator().manual_seed(1))))
for odom in ("icp", "gradicp"):
    t0 = time.time()
    pcs, poses, counts = run_reference(odom, d)
    _, poses_p, counts_p = run_reference(odom, d_pert)
    name = "pf_" + odom
    S[name + "_sens_pose"] = npy((poses_p - poses).abs().amax((0, 2, 3)) / poses.abs().amax((0, 2, 3)))
    S[name + "_sens_counts"] = np.abs(np.array(counts_p, dtype=np.int64) - np.array(counts, dtype=np.int64))
    print(name, "own sensitivity: pose", float(S[name + "_sens_pose"].max()), "map size", int(S[name + "_sens_counts"].max()), flush=True)
    S[name + "_poses"] = npy(poses)
    S[name + "_counts"] = np.array(counts, dtype=np.int64)
    for attr, lst in (("points", pcs.points_list), ("normals", pcs.normals_list), ("colors", pcs.colors_list),
                      ("feats", pcs.features_list)):
        a = npy(lst[0])
        S[f"{name}_map_{attr}"] = a[::STRIDE]
        S[f"{name}_map_{attr}_sum"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
    print(name, "done: map", counts[-1], "pose err vs gt",
          float((poses - P).abs().max()), "%.0f s" % (time.time() - t0), flush=True)
np.savez_compressed(os.path.join(OUT, "ref_slam_c3.npz"), **S)
print("ref_slam_c3.npz", os.path.getsize(os.path.join(OUT, "ref_slam_c3.npz")) // 1024, "KiB")
