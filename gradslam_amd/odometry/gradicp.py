"""GradICPOdometryProvider (reference odometry/gradicp.py:11-122)."""
from typing import Union

import torch

from ..structures.pointclouds import Pointclouds
from .base import OdometryProvider
from .icp import _check_provide_args
from .icputils import point_to_plane_gradICP

__all__ = ["GradICPOdometryProvider"]


class GradICPOdometryProvider(OdometryProvider):
    """Point-to-plane ICP with the differentiable gradLM solver."""

    def __init__(self, numiters: int = 20, damp: float = 1e-8, dist_thresh: Union[float, int, None] = None,
                 lambda_max: Union[float, int] = 2.0, B: Union[float, int] = 1.0, B2: Union[float, int] = 1.0,
                 nu: Union[float, int] = 200.0):
        self.numiters = numiters
        self.damp = damp
        self.dist_thresh = dist_thresh
        self.lambda_max = lambda_max
        self.B = B
        self.B2 = B2
        self.nu = nu

    def provide(self, maps_pointclouds: Pointclouds, frames_pointclouds: Pointclouds) -> torch.Tensor:
        _check_provide_args(maps_pointclouds, frames_pointclouds, "GradICPOdometryProvider")
        init = torch.eye(4, device=maps_pointclouds.device)
        out = []
        for b in range(len(maps_pointclouds)):
            T, _ = point_to_plane_gradICP(frames_pointclouds.points_list[b].unsqueeze(0),
                                          maps_pointclouds.points_list[b].unsqueeze(0),
                                          maps_pointclouds.normals_list[b].unsqueeze(0), init, numiters=self.numiters,
                                          damp=self.damp, dist_thresh=self.dist_thresh, lambda_max=self.lambda_max,
                                          B=self.B, B2=self.B2, nu=self.nu)
            out.append(T)
        return torch.stack(out).unsqueeze(1)
