"""ICPOdometryProvider (reference odometry/icp.py:11-97)."""
from typing import Union

import torch

from ..structures.pointclouds import Pointclouds
from .base import OdometryProvider
from .icputils import point_to_plane_ICP

__all__ = ["ICPOdometryProvider"]


def _check_provide_args(maps_pointclouds, frames_pointclouds, who: str):
    if not isinstance(maps_pointclouds, Pointclouds):
        raise TypeError("Expected maps_pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(type(maps_pointclouds)))
    if not isinstance(frames_pointclouds, Pointclouds):
        raise TypeError("Expected frames_pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(type(frames_pointclouds)))
    if maps_pointclouds.normals_list is None:
        raise ValueError("maps_pointclouds missing normals. Map normals must be provided if using {}".format(who))
    if len(maps_pointclouds) != len(frames_pointclouds):
        raise ValueError("Batch size of maps_pointclouds and frames_pointclouds should be equal ({0} != {1})".format(
            len(maps_pointclouds), len(frames_pointclouds)))


class ICPOdometryProvider(OdometryProvider):
    """Point-to-plane ICP with an LM solver; `provide` returns the transform that moves each frame
    cloud onto its map cloud."""

    def __init__(self, numiters: int = 20, damp: float = 1e-8, dist_thresh: Union[float, int, None] = None):
        self.numiters = numiters
        self.damp = damp
        self.dist_thresh = dist_thresh

    def provide(self, maps_pointclouds: Pointclouds, frames_pointclouds: Pointclouds) -> torch.Tensor:
        _check_provide_args(maps_pointclouds, frames_pointclouds, "ICPOdometryProvider")
        init = torch.eye(4, device=maps_pointclouds.device)
        out = []
        for b in range(len(maps_pointclouds)):  # sequences are independent: one device loop each
            T, _ = point_to_plane_ICP(frames_pointclouds.points_list[b].unsqueeze(0),
                                      maps_pointclouds.points_list[b].unsqueeze(0),
                                      maps_pointclouds.normals_list[b].unsqueeze(0), init, numiters=self.numiters,
                                      damp=self.damp, dist_thresh=self.dist_thresh)
            out.append(T)
        return torch.stack(out).unsqueeze(1)
