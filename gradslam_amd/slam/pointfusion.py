"""PointFusion: ICPSLAM whose mapping step is the PointFusion update
(reference slam/pointfusion.py:16-112)."""
import math
import warnings
from typing import Union

import torch

from ..structures.pointclouds import Pointclouds
from ..structures.rgbdimages import RGBDImages
from .fusionutils import update_map_fusion
from .icpslam import ICPSLAM

__all__ = ["PointFusion"]


class PointFusion(ICPSLAM):
    def __init__(self, *, odom: str = "gradicp", dist_th: Union[float, int] = 0.05, angle_th: Union[float, int] = 20,
                 sigma: Union[float, int] = 0.6, dsratio: int = 4, numiters: int = 20, damp: float = 1e-8,
                 dist_thresh: Union[float, int, None] = None, lambda_max: Union[float, int] = 2.0,
                 B: Union[float, int] = 1.0, B2: Union[float, int] = 1.0, nu: Union[float, int] = 200.0,
                 device: Union[torch.device, str, None] = None):
        super().__init__(odom=odom, dsratio=dsratio, numiters=numiters, damp=damp, dist_thresh=dist_thresh,
                         lambda_max=lambda_max, B=B, B2=B2, nu=nu, device=device)
        if not isinstance(dist_th, (float, int)):
            raise TypeError("Distance threshold must be of type float or int; but was of type {}.".format(type(dist_th)))
        if not isinstance(angle_th, (float, int)):
            raise TypeError("Angle threshold must be of type float or int; but was of type {}.".format(type(angle_th)))
        if dist_th < 0:
            warnings.warn("Distance threshold ({}) should be non-negative.".format(dist_th))
        if not ((0 <= angle_th) and (angle_th <= 90)):
            warnings.warn("Angle threshold ({}) should be non-negative and <=90.".format(angle_th))
        self.dist_th = dist_th
        rad_th = (angle_th * math.pi) / 180
        self.dot_th = torch.cos(rad_th) if torch.is_tensor(rad_th) else math.cos(rad_th)
        self.sigma = sigma

    def _map(self, pointclouds: Pointclouds, live_frame: RGBDImages, inplace: bool = False):
        fused = self._map_on_arena(pointclouds, live_frame, inplace) if self.fused_map else None
        if fused is not None:
            return fused
        return update_map_fusion(pointclouds, live_frame, self.dist_th, self.dot_th, self.sigma, inplace)

    _map._gs_arena_form = True  # _arena_update below is this mapping step on arena storage

    # arena-backed sequence driver (ICPSLAM._forward_streamed): the PointFusion update and its warnings
    _arena_features = True

    def _arena_update(self, arena, depth_s, rgb_s, K, pose, bound, stats_row):
        from .. import ops

        mp, mn, mc, mf = arena.rows(bound)
        ops.pointfusion_update_raw(depth_s, rgb_s, K, pose, mp, mn, mc, mf, arena.counts, self.dist_th, self.dot_th, self.sigma,
                                   stats_row)

    def _stream_warnings(self, s, row):
        if s == 0:
            return
        if row[0] == 0:
            warnings.warn("No active map points were found")
            return
        md = torch.tensor(row[3], dtype=torch.int32).view(torch.float32).item()
        if md > 1.001:
            warnings.warn("Max of dot product was {0} > 1. Inputs were not normalized along dim ({1}). Was this "
                          "intentional?".format(md, -1), RuntimeWarning)
        if row[1] == 0:
            warnings.warn("No similar map points were found (despite total {0} active points across the batch)".format(row[0]),
                          RuntimeWarning)
