"""Frame loop of ICPSLAM / PointFusion on plain tensors (fp32 torch CPU ops)."""
import math

import torch

from .cloud import Cloud
from .fusion import find_active_map_points, make_frame, update_map_aggregate, update_map_fusion
from .geometry import compose_transformations
from .icp import downsample_frame, downsample_map, provide


def localize(cloud: Cloud, live: dict, prev: dict, odom: str, dsratio: int, **icp_kw) -> torch.Tensor:
    """Pose of the live frame from ICP against the active, downsampled map seen from the
    previous frame.  ``live`` must already carry prev's pose.  reference slam/icpslam.py:238-247."""
    frames_pc = downsample_frame(live["gV"], live["gN"], live["rgb"], live["depth"], dsratio)
    table = find_active_map_points(cloud, prev)
    maps_pc = downsample_map(cloud, table, dsratio)
    T = provide(maps_pc, frames_pc, odom=odom, **icp_kw)
    return compose_transformations(T.squeeze(1), prev["pose"].squeeze(1)).unsqueeze(1)


def run(rgb, depth, K, poses, *, mode: str = "pointfusion", odom: str = "gradicp", dsratio: int = 4,
        numiters: int = 20, damp: float = 1e-8, dist_thresh=None, lambda_max=2.0, B=1.0, B2=1.0,
        nu=200.0, dist_th=0.05, angle_th=20, sigma=0.6, counts_out=None):
    """rgb (B,L,H,W,3), depth (B,L,H,W,1), K (B,1,4,4), poses (B,L,4,4) or None ->
    (Cloud, recovered poses (B,L,4,4)).  reference slam/icpslam.py:99-178,
    slam/pointfusion.py:102-112."""
    dot_th = math.cos((angle_th * math.pi) / 180)
    icp_kw = dict(numiters=numiters, damp=damp, dist_thresh=dist_thresh, lambda_max=lambda_max,
                  B=B, B2=B2, nu=nu)
    Bn, L = rgb.shape[:2]
    cloud = Cloud()
    out = []
    prev = None
    for s in range(L):
        sl = slice(s, s + 1)
        pose = None if poses is None else poses[:, sl]
        if s == 0 and pose is None:
            pose = torch.eye(4, dtype=torch.float).view(1, 1, 4, 4).repeat(Bn, 1, 1, 1)
        if prev is not None and odom != "gt":
            live = make_frame(rgb[:, sl], depth[:, sl], K, prev["pose"])
            pose = localize(cloud, live, prev, odom, dsratio, **icp_kw)
        live = make_frame(rgb[:, sl], depth[:, sl], K, pose)
        if mode == "pointfusion":
            cloud = update_map_fusion(cloud, live, dist_th, dot_th, sigma)
        else:
            cloud = update_map_aggregate(cloud, live)
        prev = live if odom != "gt" else None
        if counts_out is not None:  # map size of the first sequence after every frame (long-sequence goldens)
            counts_out.append(cloud.counts[0])
        out.append(pose[:, 0])
    return cloud, torch.stack(out, 1)
