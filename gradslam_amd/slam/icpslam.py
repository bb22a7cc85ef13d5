"""ICPSLAM: frame loop = localise (ICP against the active map) + map (aggregate)
(reference slam/icpslam.py:18-264)."""
import warnings
from typing import Optional, Union

import torch
import torch.nn as nn

from ..geometry.geometryutils import compose_transformations
from ..odometry.gradicp import GradICPOdometryProvider
from ..odometry.icp import ICPOdometryProvider
from ..odometry.icputils import _gather_by_table, downsample_rgbdimages
from ..structures.pointclouds import Pointclouds
from ..structures.rgbdimages import RGBDImages
from .fusionutils import _project, update_map_aggregate

__all__ = ["ICPSLAM"]


class ICPSLAM(nn.Module):
    """Point-based SLAM with ICP odometry.  Keyword arguments, defaults, `forward` / `step` contract,
    warnings and errors follow the reference class."""

    # differentiable localisation as one autograd node (gs_slam_localize_taped); False = the staged
    # formulation with one node per op, kept as an independent check of the fused reverse pass
    fused_autograd = True

    def __init__(self, *, odom: str = "gradicp", dsratio: int = 4, numiters: int = 20, damp: float = 1e-8,
                 dist_thresh: Union[float, int, None] = None, lambda_max: Union[float, int] = 2.0,
                 B: Union[float, int] = 1.0, B2: Union[float, int] = 1.0, nu: Union[float, int] = 200.0,
                 device: Union[torch.device, str, None] = None):
        super().__init__()
        if odom not in ["gt", "icp", "gradicp"]:
            msg = "odometry method ({}) not supported for PointFusion. ".format(odom)
            msg += "Currently supported odometry modules for PointFusion are: 'gt', 'icp', 'gradicp'"
            raise ValueError(msg)
        odomprov = None
        if odom == "icp":
            odomprov = ICPOdometryProvider(numiters, damp, dist_thresh)
        elif odom == "gradicp":
            odomprov = GradICPOdometryProvider(numiters, damp, dist_thresh, lambda_max, B, B2, nu)
        self.odom = odom
        self.odomprov = odomprov
        self.dsratio = dsratio
        device = torch.device(device) if device is not None else torch.device("cpu")
        self.device = torch.Tensor().to(device).device

    def forward(self, frames: RGBDImages):
        """frames (B, L, ...) -> (global map Pointclouds, recovered poses (B, L, 4, 4))."""
        if not isinstance(frames, RGBDImages):
            raise TypeError("Expected frames to be of type gradslam.RGBDImages. Got {0}.".format(type(frames)))
        pointclouds = Pointclouds(device=self.device)
        batch_size, seq_len = frames.shape[:2]
        recovered_poses = torch.empty(batch_size, seq_len, 4, 4).to(self.device)
        prev_frame = None
        for s in range(seq_len):  # true serial dependence: pose s needs map s-1
            live_frame = frames[:, s].to(self.device)
            if s == 0 and live_frame.poses is None:
                live_frame.poses = torch.eye(4, dtype=torch.float, device=self.device).view(1, 1, 4, 4).repeat(
                    batch_size, 1, 1, 1)
            pointclouds, live_frame.poses = self.step(pointclouds, live_frame, prev_frame, inplace=True)
            prev_frame = live_frame if self.odom != "gt" else None
            recovered_poses[:, s] = live_frame.poses[:, 0]
        return pointclouds, recovered_poses

    def step(self, pointclouds: Pointclouds, live_frame: RGBDImages, prev_frame: Optional[RGBDImages] = None,
             inplace: bool = False):
        """One SLAM step on `live_frame` (sequence length 1) -> (updated map, live poses (B,1,4,4))."""
        if not isinstance(live_frame, RGBDImages):
            raise TypeError("Expected live_frame to be of type gradslam.RGBDImages. Got {0}.".format(type(live_frame)))
        live_frame.poses = self._localize(pointclouds, live_frame, prev_frame)
        pointclouds = self._map(pointclouds, live_frame, inplace)
        return pointclouds, live_frame.poses

    def _localize(self, pointclouds: Pointclouds, live_frame: RGBDImages, prev_frame: RGBDImages):
        if not isinstance(pointclouds, Pointclouds):
            raise TypeError("Expected pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(type(pointclouds)))
        if not isinstance(live_frame, RGBDImages):
            raise TypeError("Expected live_frame to be of type gradslam.RGBDImages. Got {0}.".format(type(live_frame)))
        if not isinstance(prev_frame, (RGBDImages, type(None))):
            raise TypeError("Expected prev_frame to be of type gradslam.RGBDImages or None. Got {0}.".format(type(prev_frame)))
        if prev_frame is not None:
            if self.odom == "gt":
                warnings.warn("`prev_frame` is not used when using `odom='gt'` (should be None)")
            elif not prev_frame.has_poses:
                raise ValueError("`prev_frame` should have poses, but did not.")
        if prev_frame is None and pointclouds.has_points and self.odom != "gt":
            warnings.warn("`prev_frame` was None despite `{}` odometry method. Using `live_frame` poses.".format(self.odom))
        if prev_frame is None or self.odom == "gt":
            if not live_frame.has_poses:
                raise ValueError("`live_frame` must have poses when `prev_frame` is None or `odom='gt'`.")
            return live_frame.poses

        if self.odom in ["icp", "gradicp"]:
            fused = self._localize_fused(pointclouds, live_frame, prev_frame)
            if fused is not None:
                return fused
            live_frame.poses = prev_frame.poses
            frames_pc = downsample_rgbdimages(live_frame, self.dsratio)
            # active-point search with the ds-grid filter fused in (find_active_map_points +
            # downsample_pointclouds' row filter of the reference, one pass over the map)
            rows, cnt = _project(pointclouds, prev_frame, self.dsratio)
            maps_pc = _gather_by_table(pointclouds, rows[: int(cnt.item())], len(pointclouds))
            transform = self.odomprov.provide(maps_pc, frames_pc)
            return compose_transformations(transform.squeeze(1), prev_frame.poses.squeeze(1)).unsqueeze(1)

    def _localize_fused(self, pointclouds: Pointclouds, live_frame: RGBDImages, prev_frame: RGBDImages):
        """Sync-free single-call localisation (`gs_slam_localize`) when nothing needs gradients.  Same
        results as the staged path below it; the staged path stays for autograd."""
        from .. import ops

        p = self.odomprov
        tensors = (live_frame.depth_image, live_frame.intrinsics, prev_frame.poses)
        if (not pointclouds.has_points or not pointclouds.has_normals or live_frame.channels_first
                or not all(t.is_cuda for t in tensors) or len(pointclouds) != len(live_frame)):
            return None
        mp, mn = pointclouds.points_padded, pointclouds.normals_padded
        gparams = (p.lambda_max, p.B, p.B2, p.nu) if self.odom == "gradicp" else None
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors + (mp, mn)):
            if not self.fused_autograd:
                return None  # staged path below: one autograd node per op, like the reference's graph
            # one autograd node for the whole step; the live frame's maps under the previous pose stay an
            # ordinary differentiable input so that their adjoint reaches depth / intrinsics / pose
            live_frame.poses = prev_frame.poses
            return ops.slam_localize_autograd(live_frame.global_vertex_map, live_frame.depth_image, live_frame.intrinsics,
                                              prev_frame.poses, mp, mn, pointclouds._counts_i32(), self.dsratio, p.numiters,
                                              p.damp, p.dist_thresh, gparams)
        poses, V, N = ops.slam_localize_raw(live_frame.depth_image, live_frame.intrinsics, prev_frame.poses, mp, mn,
                                            pointclouds._counts_i32(), self.dsratio, p.numiters, p.damp, p.dist_thresh,
                                            gparams)
        live_frame._poses = prev_frame.poses  # what the reference leaves behind (:239)
        live_frame._vertex_map, live_frame._normal_map = V, N  # pose-independent: reusable by _map
        live_frame._global_vertex_map = live_frame._global_normal_map = None
        return poses

    def _map(self, pointclouds: Pointclouds, live_frame: RGBDImages, inplace: bool = False):
        return update_map_aggregate(pointclouds, live_frame, inplace)
