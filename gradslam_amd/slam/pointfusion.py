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

    def _can_fuse_sequence(self, frames) -> bool:
        """forward() with gradients as one node: channels-last float32 sequences on the device, the mapping step not
        overridden by a subclass."""
        if not getattr(type(self)._map, "_gs_arena_form", False) or frames.channels_first or self.device.type != "cuda":
            return False
        tensors = (frames.rgb_image, frames.depth_image, frames.intrinsics, frames.poses)
        if frames.poses is None and self.odom == "gt":
            return False
        if any(t is not None and (not t.is_cuda or t.dtype != torch.float32) for t in tensors):
            return False
        if not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)):
            return False  # nothing to differentiate: the streamed / step-by-step drivers
        if not (frames.shape[0] <= 60 and frames.shape[2] >= 2 and frames.shape[3] >= 2):
            return False
        # Memory law of the sequence node: it keeps, until backward, one fusion tape (44 B per pixel) and one localisation
        # tape (the ICP loop's clouds and neighbour arrays, ~20 B per ds-grid point and association, + 4 B per target slot)
        # per frame, and the arena -- ~7.7 GB for 200 frames of 640x480.  A sequence whose tapes would not fit takes the
        # per-frame nodes instead (slower, but every frame's intermediates are freed as autograd walks back).
        need = frames.shape[0] * self.sequence_tape_bytes(frames.shape[1], frames.shape[2], frames.shape[3])
        free = torch.cuda.mem_get_info(frames.device)[0]
        if need > 0.8 * free:
            warnings.warn("PointFusion: the one-node differentiable sequence would keep ~{:.1f} GB of tapes ({:.1f} GB free): "
                          "taking one autograd node per frame instead".format(need / 1e9, free / 1e9))
            return False
        return True

    def sequence_tape_bytes(self, L: int, H: int, W: int) -> int:
        """Projected bytes the one-node differentiable sequence keeps until backward (see _can_fuse_sequence): exact in the
        per-pixel and per-iteration terms, with the map's growth taken as 5 % of a frame per frame (what the synthetic and
        the reference's fixture sequences show; the target-slot term is the only one that depends on it)."""
        p = self.odomprov
        iters = p.numiters if p is not None else 0
        ds = max(int(self.dsratio), 1)
        src = -(-H // ds) * -(-W // ds)
        assoc = (2 * iters if self.odom == "gradicp" else iters + 1) if self.odom != "gt" else 0
        map_pts = int((1.0 + 0.05 * L) * H * W)
        cap_t = 1 << max(map_pts - 1, 1).bit_length()
        per_frame = 44 * H * W + assoc * 20 * src + 16 * src + (4 * cap_t if assoc else 0)
        return L * per_frame + 2 * 40 * map_pts  # + the arena and its copy in backward

    def _forward_sequence_node(self, frames: RGBDImages):
        from .. import ops
        from .icpslam import _MapArena

        p = self.odomprov
        gparams = (p.lambda_max, p.B, p.B2, p.nu) if self.odom == "gradicp" else None
        cfg = (self.odom, self.dsratio, p.numiters if p is not None else 0, p.damp if p is not None else 0.0,
               p.dist_thresh if p is not None else None, gparams, self.dist_th, self.dot_th, self.sigma, _MapArena)
        pts, nrm, col, cc, poses, stats = ops.pointfusion_sequence_autograd(frames.rgb_image, frames.depth_image, frames.intrinsics,
                                                                            frames.poses, cfg)
        for s, row in enumerate(stats.tolist()):  # the reference's warnings, raised once the sequence is done
            if row[2]:
                raise RuntimeError("map arena overflow at frame {} (internal capacity bound violated)".format(s))
            self._stream_warnings(s, row)
        n = [int(x) for x in stats[:, 4:4 + pts.shape[0]].sum(0).tolist()]  # rows every sequence's map holds (appended, summed)
        pick = lambda x: [x[b, : n[b]] for b in range(x.shape[0])]
        return Pointclouds(points=pick(pts), normals=pick(nrm), colors=pick(col), features=pick(cc)), poses

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
