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


class _MapArena:
    """Device-resident map storage for the streamed sequence driver: capacity-doubling
    (B, cap, C) arrays, per-sequence point counts that live on the device, O(1) append.  The host only tracks an
    upper bound of the counts; it is tightened by asynchronous read-backs (pinned memory + events) that never
    stall the frame loop."""

    def __init__(self, B: int, hw: int, device):
        self.B, self.hw, self.device = B, hw, device
        self.cap = 1 << max(2 * hw - 1, 1).bit_length()
        mk = lambda c: torch.zeros((B, self.cap, c), dtype=torch.float32, device=device)
        self.points, self.normals, self.colors, self.ccounts = mk(3), mk(3), mk(3), mk(1)
        self.counts = torch.zeros(B, dtype=torch.int32, device=device)
        self.upper = 0          # host-side upper bound of max_b counts[b]
        self.appends = 0        # frames appended so far
        self._pinned = [torch.empty(B, dtype=torch.int32).pin_memory() for _ in range(4)]
        self._pending = []      # (appends at issue time, pinned buffer, event)

    def _tighten(self):
        while self._pending and self._pending[0][2].query():
            at, buf, _ = self._pending.pop(0)
            self._pinned.append(buf)
            self.upper = min(self.upper, int(buf.max()) + (self.appends - at) * self.hw)

    def reserve_frame(self) -> int:
        """Make room for one more frame's worth of rows; returns the row bound to hand to the kernels."""
        self._tighten()
        need = self.upper + self.hw
        if need > self.cap:
            new_cap = 1 << (need - 1).bit_length()
            for name in ("points", "normals", "colors", "ccounts"):
                old = getattr(self, name)
                new = torch.zeros((self.B, new_cap, old.shape[2]), dtype=torch.float32, device=self.device)
                new[:, : self.cap] = old
                setattr(self, name, new)
            self.cap = new_cap
        # one sequence: any row bound works as the "padded length" (rows beyond the count are zero); several
        # sequences share the row stride, which is the capacity
        return need if self.B == 1 else self.cap

    def rows(self, bound: int):
        """(points, normals, colors, ccounts) views of `bound` rows per sequence (contiguous)."""
        if bound == self.cap:
            return self.points, self.normals, self.colors, self.ccounts
        return self.points[:, :bound], self.normals[:, :bound], self.colors[:, :bound], self.ccounts[:, :bound]

    def appended(self):
        self.upper += self.hw
        self.appends += 1
        if self._pinned:
            buf = self._pinned.pop()
            buf.copy_(self.counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((self.appends, buf, ev))

    def to_pointclouds(self) -> Pointclouds:
        n = self.counts.tolist()  # the one host synchronisation of a streamed sequence
        pick = lambda x: [x[b, : n[b]].clone() for b in range(self.B)]
        return Pointclouds(points=pick(self.points), normals=pick(self.normals), colors=pick(self.colors),
                           features=pick(self.ccounts))


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
        return update_map_fusion(pointclouds, live_frame, self.dist_th, self.dot_th, self.sigma, inplace)

    # whole sequences without gradients run on the arena-backed driver: two C calls per frame, no host
    # synchronisation until the map is handed back.  Same results as the step-by-step path (same kernels).
    streamed = True

    def forward(self, frames: RGBDImages):
        if self.streamed and self._can_stream(frames):
            return self._forward_streamed(frames)
        return super().forward(frames)

    def _can_stream(self, frames) -> bool:
        if not isinstance(frames, RGBDImages) or frames.channels_first or self.device.type != "cuda":
            return False
        tensors = (frames.rgb_image, frames.depth_image, frames.intrinsics, frames.poses)
        if frames.poses is None and self.odom == "gt":
            return False  # the step-by-step path raises the reference's error for this
        if any(t is not None and (not t.is_cuda or t.dtype != torch.float32) for t in tensors):
            return False
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
            return False
        return frames.shape[0] <= 60 and frames.shape[2] >= 2 and frames.shape[3] >= 2

    def _forward_streamed(self, frames: RGBDImages):
        from .. import ops

        B, L, H, W = frames.shape
        dev = frames.device
        rgb, depth, K = frames.rgb_image.detach(), frames.depth_image.detach(), frames.intrinsics.detach().contiguous()
        gt_poses = frames.poses.detach() if frames.poses is not None else None
        arena = _MapArena(B, H * W, dev)
        recovered = torch.empty((B, L, 4, 4), dtype=torch.float32, device=dev)
        stats = torch.zeros((L, 4 + B), dtype=torch.int32, device=dev)
        p = self.odomprov
        gparams = (p.lambda_max, p.B, p.B2, p.nu) if self.odom == "gradicp" else None
        frame = lambda x, s: x[:, s].contiguous()  # (B,H,W,C); a view (no copy) for one contiguous sequence
        prev = None
        for s in range(L):  # true serial dependence: pose s needs map s-1
            d_s, c_s = frame(depth, s), frame(rgb, s)
            if s == 0 or self.odom == "gt":
                pose = (gt_poses[:, s:s + 1] if gt_poses is not None else
                        torch.eye(4, dtype=torch.float32, device=dev).view(1, 1, 4, 4).repeat(B, 1, 1, 1))
            else:
                mp, mn, _, _ = arena.rows(bound)
                pose, _, _ = ops.slam_localize_raw(d_s.unsqueeze(1), K, prev, mp, mn, arena.counts, self.dsratio, p.numiters,
                                                   p.damp, p.dist_thresh, gparams)
            bound = arena.reserve_frame()
            mp, mn, mc, mf = arena.rows(bound)
            ops.pointfusion_update_raw(d_s, c_s, K, pose, mp, mn, mc, mf, arena.counts, self.dist_th, self.dot_th, self.sigma,
                                       stats[s])
            arena.appended()
            recovered[:, s] = pose[:, 0]
            prev = pose
        pointclouds = arena.to_pointclouds()
        for s, row in enumerate(stats.tolist()):  # the reference's warnings, raised once the sequence is done
            if row[2]:
                raise RuntimeError("PointFusion arena overflow at frame {} (internal capacity bound violated)".format(s))
            if s == 0:
                continue
            if row[0] == 0:
                warnings.warn("No active map points were found")
                continue
            md = torch.tensor(row[3], dtype=torch.int32).view(torch.float32).item()
            if md > 1.001:
                warnings.warn("Max of dot product was {0} > 1. Inputs were not normalized along dim ({1}). Was this "
                              "intentional?".format(md, -1), RuntimeWarning)
            if row[1] == 0:
                warnings.warn("No similar map points were found (despite total {0} active points across the batch)".format(
                    row[0]), RuntimeWarning)
        return pointclouds, recovered
