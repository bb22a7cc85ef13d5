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


import os as _os

_NO_READBACK = bool(_os.environ.get("GS_NO_READBACK"))  # measurements only: never tighten the host's bound of the map size


class _MapArena:
    """Device-resident map storage for the streamed sequence driver: capacity-doubling
    (B, cap, C) arrays, per-sequence point counts that live on the device, O(1) append.  The host only tracks an
    upper bound of the counts; it is tightened by asynchronous read-backs (pinned memory + events) that never
    stall the frame loop."""

    def __init__(self, B: int, hw: int, device, with_features: bool = True):
        self.B, self.hw, self.device = B, hw, device
        self.cap = 1 << max(2 * hw - 1, 1).bit_length()
        mk = lambda c: torch.zeros((B, self.cap, c), dtype=torch.float32, device=device)
        self.points, self.normals, self.colors = mk(3), mk(3), mk(3)
        self.ccounts = mk(1) if with_features else None
        self.counts = torch.zeros(B, dtype=torch.int32, device=device)
        self.upper = 0          # host-side upper bound of max_b counts[b]
        self.appends = 0        # frames appended so far
        self._pinned = [torch.empty(B, dtype=torch.int32).pin_memory() for _ in range(4)]
        self._pending = []      # (appends at issue time, pinned buffer, event)

    @classmethod
    def for_one_step(cls, pointclouds: Pointclouds, B: int, hw: int, device, with_features: bool):
        """Arena holding a copy of `pointclouds` with room for exactly one more frame (the step-by-step API)."""
        self = cls.__new__(cls)
        self.B, self.hw, self.device = B, hw, device
        n_old = max(pointclouds._counts) if pointclouds.has_points else 0
        self.cap = n_old + hw
        mk = lambda c: torch.zeros((B, self.cap, c), dtype=torch.float32, device=device)
        self.points, self.normals, self.colors = mk(3), mk(3), mk(3)
        self.ccounts = mk(1) if with_features else None
        if pointclouds.has_points:
            self.points[:, :n_old] = pointclouds.points_padded
            self.normals[:, :n_old] = pointclouds.normals_padded
            self.colors[:, :n_old] = pointclouds.colors_padded
            if with_features:
                self.ccounts[:, :n_old] = pointclouds.features_padded
            self.counts = pointclouds._counts_i32().clone()
        else:
            self.counts = torch.zeros(B, dtype=torch.int32, device=device)
        self.upper, self.appends, self._pinned, self._pending = n_old, 0, [], []
        return self

    # The host may run at most this many frames ahead of the map counts it knows.  The row bound it hands to the kernels
    # and the capacity it reserves are `last known count + frames since x H W`: a host that enqueues four times faster than
    # the GPU executes (0.13 against 0.4-0.6 ms per frame on the round-3 boxes) would otherwise be ~150 frames ahead by
    # the end of a 200-frame sequence, none of its read-backs would have landed when it needs them, and the arena would be
    # grown -- allocated, zero-filled, copied -- as if every frame had appended H W points (46 M rows instead of 3.3 M:
    # the 200-frame forward ran at 1 600 frames/s where the same kernels with exact bounds run at 2 650, and FASTER under
    # rocprofv3, whose per-launch host cost happens to keep the host near the GPU: tools/gap_bisect.py).  Two frames of
    # queued work (~1 ms) keep the GPU fed on any host.
    MAX_FRAMES_AHEAD = 2

    def _tighten(self):
        while self._pending and (self._pending[0][2].query() or self.appends - self._pending[0][0] >= self.MAX_FRAMES_AHEAD):
            at, buf, ev = self._pending.pop(0)
            ev.synchronize()  # (returns at once when the query above succeeded)
            self._pinned.append(buf)
            self.upper = min(self.upper, int(buf.max()) + (self.appends - at) * self.hw)

    def reserve_frame(self) -> int:
        """Make room for one more frame's worth of rows; returns the row bound to hand to the kernels."""
        self._tighten()
        need = self.upper + self.hw
        if need > self.cap:
            # one sequence: power-of-two capacities keep the workspace ADDRESS (and with it the captured graph) for many
            # frames, and the kernels get `need`, not the capacity, as row bound.  Several sequences share the row
            # stride = capacity, which the kernels then scan: grow by 1.5x in 64 Ki-row steps instead, so that the
            # padding they read stays below half of the data
            new_cap = 1 << (need - 1).bit_length() if self.B == 1 else -(-max(need, self.cap + self.cap // 2) // 65536) * 65536
            for name in ("points", "normals", "colors", "ccounts"):
                old = getattr(self, name)
                if old is None:
                    continue
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
        cut = lambda x: None if x is None else x[:, :bound]
        return cut(self.points), cut(self.normals), cut(self.colors), cut(self.ccounts)

    def appended(self):
        self.upper += self.hw
        self.appends += 1
        if self._pinned and not _NO_READBACK:
            buf = self._pinned.pop()
            buf.copy_(self.counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((self.appends, buf, ev))

    def to_pointclouds(self) -> Pointclouds:
        n = self.counts.tolist()  # the one host synchronisation of a streamed sequence
        pick = lambda x: [x[b, : n[b]].clone() for b in range(self.B)]
        return Pointclouds(points=pick(self.points), normals=pick(self.normals), colors=pick(self.colors),
                           features=pick(self.ccounts) if self.ccounts is not None else None)



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

    # Whole sequences without gradients run on the arena-backed driver: two C calls per frame (localise, map
    # update), map counts resident on the device, no host synchronisation until the map is handed back.  Same
    # results as the step-by-step path (same kernels); `streamed = False` forces the latter.
    streamed = True
    fused_map = True   # step(): the mapping step as one fused call when nothing needs gradients (False = staged)
    # forward() WITH gradients as one autograd node per sequence (PointFusion, one sequence per call); False = one node
    # per op / per localisation, kept as the independent check of the fused reverse pass
    fused_sequence_autograd = True

    def _can_fuse_sequence(self, frames) -> bool:
        return False  # ICPSLAM's aggregate map keeps the per-frame nodes (PointFusion overrides)
    _arena_features = False

    def forward(self, frames: RGBDImages):
        """frames (B, L, ...) -> (global map Pointclouds, recovered poses (B, L, 4, 4))."""
        if not isinstance(frames, RGBDImages):
            raise TypeError("Expected frames to be of type gradslam.RGBDImages. Got {0}.".format(type(frames)))
        if self.streamed and self._can_stream(frames):
            return self._forward_streamed(frames)
        if self.fused_sequence_autograd and self._can_fuse_sequence(frames):
            return self._forward_sequence_node(frames)
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

    def _can_stream(self, frames) -> bool:
        # a subclass with a mapping step of its own keeps the step-by-step loop
        if not getattr(type(self)._map, "_gs_arena_form", False):
            return False
        if frames.channels_first or self.device.type != "cuda":
            return False
        tensors = (frames.rgb_image, frames.depth_image, frames.intrinsics, frames.poses)
        if frames.poses is None and self.odom == "gt":
            return False  # the step-by-step path raises the reference's error for this
        if any(t is not None and (not t.is_cuda or t.dtype != torch.float32) for t in tensors):
            return False
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
            return False
        return frames.shape[0] <= 60 and frames.shape[2] >= 2 and frames.shape[3] >= 2

    def _arena_update(self, arena, depth_s, rgb_s, K, pose, bound, stats_row):
        """update_map_aggregate on the arena (PointFusion overrides this with the fusion update)."""
        from .. import ops

        mp, mn, mc, _ = arena.rows(bound)
        ops.aggregate_update_raw(depth_s, rgb_s, K, pose, mp, mn, mc, arena.counts, stats_row)

    def _stream_warnings(self, s, row):
        pass

    def _forward_streamed(self, frames: RGBDImages):
        from .. import ops

        B, L, H, W = frames.shape
        dev = frames.device
        rgb, depth, K = frames.rgb_image.detach(), frames.depth_image.detach(), frames.intrinsics.detach().contiguous()
        gt_poses = frames.poses.detach() if frames.poses is not None else None
        arena = _MapArena(B, H * W, dev, with_features=self._arena_features)
        recovered = torch.empty((B, L, 4, 4), dtype=torch.float32, device=dev)
        stats = torch.zeros((L, 4 + B), dtype=torch.int32, device=dev)
        p = self.odomprov
        gparams = (p.lambda_max, p.B, p.B2, p.nu) if self.odom == "gradicp" else None
        frame = lambda x, s: x[:, s].contiguous()  # (B,H,W,C); a view (no copy) for one contiguous sequence
        prev, bound = None, None
        for s in range(L):  # true serial dependence: pose s needs map s-1
            d_s, c_s = frame(depth, s), frame(rgb, s)
            if s == 0 or self.odom == "gt":
                pose = (gt_poses[:, s:s + 1] if gt_poses is not None else
                        torch.eye(4, dtype=torch.float32, device=dev).view(1, 1, 4, 4).repeat(B, 1, 1, 1))
            else:
                mp, mn, _, _ = arena.rows(bound)
                # (one sequence: the pose goes straight into its row of the result, and the local maps nobody reads are not
                # written -- a copy launch and 24 B per pixel less per frame)
                pose, _, _ = ops.slam_localize_raw(d_s.unsqueeze(1), K, prev, mp, mn, arena.counts, self.dsratio, p.numiters,
                                                   p.damp, p.dist_thresh, gparams, out=recovered[:, s:s + 1] if B == 1 else None,
                                                   want_maps=False)
            bound = arena.reserve_frame()
            self._arena_update(arena, d_s, c_s, K, pose, bound, stats[s])
            arena.appended()
            if pose.data_ptr() != recovered[:, s:s + 1].data_ptr():
                recovered[:, s] = pose[:, 0]
            prev = pose
        pointclouds = arena.to_pointclouds()
        rows = stats.tolist()
        # diagnostics (no reference counterpart): rows appended per frame and sequence -- their running sum is the map size
        # after every frame, which the long-sequence parity tests compare with the reference's
        self.last_appended = [row[4:4 + B] for row in rows]
        for s, row in enumerate(rows):  # the reference's warnings, raised once the sequence is done
            if row[2]:
                raise RuntimeError("map arena overflow at frame {} (internal capacity bound violated)".format(s))
            self._stream_warnings(s, row)
        return pointclouds, recovered

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

    def _map_on_arena(self, pointclouds: Pointclouds, live_frame: RGBDImages, inplace: bool):
        """The mapping step as ONE fused call + ONE host read (counts and warning counters together) when nothing
        needs gradients and the arguments are well formed; None = take the staged path (which also raises the
        reference's errors for malformed arguments).  Same kernels, same results as the staged path."""
        if not (isinstance(pointclouds, Pointclouds) and isinstance(live_frame, RGBDImages)):
            return None
        if live_frame.channels_first or live_frame.shape[1] != 1 or live_frame.poses is None:
            return None
        B, _, H, W = live_frame.shape
        tensors = (live_frame.rgb_image, live_frame.depth_image, live_frame.intrinsics, live_frame.poses)
        if any(not t.is_cuda or t.dtype != torch.float32 for t in tensors) or B > 60 or H < 2 or W < 2:
            return None
        dev = live_frame.depth_image.device
        feats = self._arena_features
        if pointclouds.has_points:
            if (len(pointclouds) != B or pointclouds.device != dev or not pointclouds.has_normals or not pointclouds.has_colors
                    or pointclouds.has_features != feats or (feats and pointclouds.num_features != 1)):
                return None
            held = (pointclouds.points_padded, pointclouds.normals_padded, pointclouds.colors_padded)
        else:
            if pointclouds.device != dev:
                return None
            held = ()
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors + held):
            return None
        had = pointclouds.has_points
        n_old = list(pointclouds._counts) if had else [0] * B
        arena = _MapArena.for_one_step(pointclouds, B, H * W, dev, feats)
        stats = torch.zeros(4 + B, dtype=torch.int32, device=dev)
        self._arena_update(arena, live_frame.depth_image[:, 0].contiguous(), live_frame.rgb_image[:, 0].contiguous(),
                           live_frame.intrinsics.contiguous(), live_frame.poses, arena.cap, stats)
        host = torch.cat([arena.counts, stats]).tolist()  # the one host synchronisation of the step
        n_new, row = host[:B], host[B:]
        if row[2]:
            raise RuntimeError("map arena overflow (internal capacity bound violated)")
        if had:
            self._stream_warnings(1, row)
        arrays = (arena.points, arena.normals, arena.colors, arena.ccounts)
        if inplace:
            return pointclouds._adopt_rows(arrays, n_new)
        if had:  # like the reference, the merged rows also land in the object that was passed in
            cut = max(n_old)
            pointclouds._adopt_rows(tuple(None if x is None else x[:, :cut].clone() for x in arrays), n_old)
        return Pointclouds(device=dev)._adopt_rows(arrays, n_new)

    def _map(self, pointclouds: Pointclouds, live_frame: RGBDImages, inplace: bool = False):
        fused = self._map_on_arena(pointclouds, live_frame, inplace) if self.fused_map else None
        return fused if fused is not None else update_map_aggregate(pointclouds, live_frame, inplace)

    _map._gs_arena_form = True  # _arena_update above is this mapping step on arena storage
