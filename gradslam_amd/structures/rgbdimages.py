"""RGBDImages: batch of RGB-D sequences with lazily computed vertex / normal maps.

Same public surface and error contracts as the reference container
(structures/rgbdimages.py:13-762).  The four lazy maps -- local and global vertex map, local
and global normal map -- are produced by ONE fused HIP launch (gradslam_amd/csrc/maps.hip,
`gs_vertex_normal_maps`) instead of the reference's chain of meshgrid / einsum / cross / norm ops,
and carry a hand-written adjoint so gradients reach depth, intrinsics and poses.
Plot helpers (plotly) are out of scope.
"""
from typing import Optional, Union

import torch

from .. import ops

__all__ = ["RGBDImages"]


def _resolve(device):
    return torch.Tensor().to(device).device


class RGBDImages(object):
    _INTERNAL_TENSORS = ["_rgb_image", "_depth_image", "_intrinsics", "_poses", "_pixel_pos", "_vertex_map",
                         "_normal_map", "_global_vertex_map", "_global_normal_map"]

    def __init__(self, rgb_image, depth_image, intrinsics, poses=None, channels_first: bool = False, device=None, *,
                 pixel_pos=None):
        super().__init__()
        for name, val, opt in (("rgb_image", rgb_image, False), ("depth_image", depth_image, False),
                               ("intrinsics", intrinsics, False), ("poses", poses, True), ("pixel_pos", pixel_pos, True)):
            if not (torch.is_tensor(val) or (opt and val is None)):
                raise TypeError("Expected {} to be of type tensor{}; got {}".format(name, " or None" if opt else "", type(val)))
        if not isinstance(channels_first, bool):
            raise TypeError("Expected channels_first to be of type bool; got {}".format(type(channels_first)))
        self._channels_first = channels_first

        if rgb_image.ndim != 5:
            raise ValueError("rgb_image should have ndim=5, but had ndim={}".format(rgb_image.ndim))
        if depth_image.ndim != 5:
            raise ValueError("depth_image should have ndim=5, but had ndim={}".format(depth_image.ndim))
        if intrinsics.ndim != 4:
            raise ValueError("intrinsics should have ndim=4, but had ndim={}".format(intrinsics.ndim))
        if poses is not None and poses.ndim != 4:
            raise ValueError("poses should have ndim=4, but had ndim={}".format(poses.ndim))

        cd = self.cdim
        self._rgb_image_shape = rgb_image.shape
        self._depth_shape = tuple(v if i != cd else 1 for i, v in enumerate(rgb_image.shape))
        self._depth_image_shape = self._depth_shape  # the reference's setter reads this name (:424)
        self._intrinsics_shape = (rgb_image.shape[0], 1, 4, 4)
        self._poses_shape = (*rgb_image.shape[:2], 4, 4)
        self._pixel_pos_shape = (*rgb_image.shape[:cd], *rgb_image.shape[cd + 1:], 3)

        if rgb_image.shape[cd] != 3:
            raise ValueError("Expected rgb_image to have 3 channels on dimension {0}. Got {1} instead".format(cd, rgb_image.shape[cd]))
        if depth_image.shape != self._depth_shape:
            raise ValueError("Expected depth_image to have shape {0}. Got {1} instead".format(self._depth_shape, depth_image.shape))
        if intrinsics.shape != self._intrinsics_shape:
            raise ValueError("Expected intrinsics to have shape {0}. Got {1} instead".format(self._intrinsics_shape, intrinsics.shape))
        if poses is not None and poses.shape != self._poses_shape:
            raise ValueError("Expected poses to have shape {0}. Got {1} instead".format(self._poses_shape, poses.shape))
        if pixel_pos is not None and pixel_pos.shape != self._pixel_pos_shape:
            raise ValueError("Expected pixel_pos to have shape {0}. Got {1} instead".format(self._pixel_pos_shape, pixel_pos.shape))

        devices = {x.device for x in (rgb_image, depth_image, intrinsics, poses, pixel_pos) if x is not None}
        if len(devices) != 1:
            raise ValueError("All inputs must be on same device, but got more than 1 device: {}".format(devices))

        self._rgb_image = rgb_image if device is None else rgb_image.to(device)
        self.device = self._rgb_image.device
        self._depth_image = depth_image.to(self.device)
        self._intrinsics = intrinsics.to(self.device)
        self._poses = poses.to(self.device) if poses is not None else None
        self._pixel_pos = pixel_pos.to(self.device) if pixel_pos is not None else None
        self._vertex_map = self._global_vertex_map = None
        self._normal_map = self._global_normal_map = None
        self._valid_depth_mask = None

        self._B, self._L = self._rgb_image.shape[:2]
        self.h = self._rgb_image.shape[3 if channels_first else 2]
        self.w = self._rgb_image.shape[4 if channels_first else 3]
        self.shape = (self._B, self._L, self.h, self.w)

    # ------------------------------------------------------------------ indexing
    def __getitem__(self, index):
        if not isinstance(index, (tuple, int)):
            raise IndexError(index)
        if isinstance(index, int):
            sl = (slice(index, index + 1), slice(None, None))
        else:
            if len(index) > 2:
                raise IndexError("Only batch and sequences can be indexed")
            sl = tuple(slice(x, x + 1) if isinstance(x, int) else x for x in index)
        new_rgb = self._rgb_image[sl[0], sl[1]]
        if new_rgb.shape[0] == 0:
            raise IndexError("Incorrect indexing at dimension 0, make sure range is within 0 and {0}".format(self._B))
        if new_rgb.shape[1] == 0:
            raise IndexError("Incorrect indexing at dimension 1, make sure range is within 0 and {0}".format(self._L))
        other = RGBDImages(new_rgb, self._depth_image[sl[0], sl[1]], self._intrinsics[sl[0], :],
                           channels_first=self.channels_first)
        for k in self._INTERNAL_TENSORS:
            if k in ("_rgb_image", "_depth_image", "_intrinsics"):
                continue
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(other, k, v[sl[0], sl[1]])
        return other

    def __len__(self):
        return self._B

    # ------------------------------------------------------------------ plain properties
    channels_first = property(lambda self: self._channels_first)
    cdim = property(lambda self: 2 if self._channels_first else 4)
    rgb_image = property(lambda self: self._rgb_image)
    depth_image = property(lambda self: self._depth_image)
    intrinsics = property(lambda self: self._intrinsics)
    poses = property(lambda self: self._poses)
    pixel_pos = property(lambda self: self._pixel_pos)
    has_poses = property(lambda self: self._poses is not None)

    @property
    def valid_depth_mask(self):
        if self._valid_depth_mask is None:
            self._valid_depth_mask = self._depth_image > 0
        return self._valid_depth_mask

    # ------------------------------------------------------------------ lazy maps (one fused launch)
    def _cl(self, t):  # to channels-last for the kernel
        return t.permute(0, 1, 3, 4, 2) if self._channels_first else t

    def _from_cl(self, t):
        if t is None:
            return None
        return t.permute(0, 1, 4, 2, 3).contiguous() if self._channels_first else t

    def _compute_maps(self, need_local: bool, need_global: bool):
        """Fill whichever of the four caches are missing.  When the poses are None the global maps
        are clones of the local ones (reference :683-685, :747-749)."""
        want_local = need_local and (self._vertex_map is None or self._normal_map is None)
        want_global = need_global and (self._global_vertex_map is None or self._global_normal_map is None)
        if need_global and self._poses is None:
            want_local = want_local or self._vertex_map is None or self._normal_map is None
            want_global = False
        if want_local or want_global:
            # computing the local pair alongside is free (same pass) and saves a second launch later
            also_local = want_local or self._vertex_map is None
            V, N, gV, gN = ops.vertex_normal_maps(self._cl(self._depth_image), self._intrinsics, self._poses,
                                                  want_local=also_local, want_global=want_global)
            if also_local:
                self._vertex_map, self._normal_map = self._from_cl(V), self._from_cl(N)
            if want_global:
                self._global_vertex_map, self._global_normal_map = self._from_cl(gV), self._from_cl(gN)
        if need_global and self._poses is None:
            if self._global_vertex_map is None:
                self._global_vertex_map = self._vertex_map.clone()
            if self._global_normal_map is None:
                self._global_normal_map = self._normal_map.clone()

    def _compute_vertex_map(self):
        self._compute_maps(True, False)

    def _compute_normal_map(self):
        self._compute_maps(True, False)

    def _compute_global_vertex_map(self):
        self._compute_maps(False, True)

    def _compute_global_normal_map(self):
        self._compute_maps(False, True)

    @property
    def vertex_map(self):
        if self._vertex_map is None:
            self._compute_vertex_map()
        return self._vertex_map

    @property
    def normal_map(self):
        if self._normal_map is None:
            self._compute_normal_map()
        return self._normal_map

    @property
    def global_vertex_map(self):
        if self._global_vertex_map is None:
            self._compute_global_vertex_map()
        return self._global_vertex_map

    @property
    def global_normal_map(self):
        if self._global_normal_map is None:
            self._compute_global_normal_map()
        return self._global_normal_map

    # ------------------------------------------------------------------ setters (cache invalidation)
    @staticmethod
    def _assert_shape(value: torch.Tensor, shape: tuple):
        if value.shape != shape:
            raise ValueError("Expected value to have shape {0}. Got {1} instead".format(shape, value.shape))

    def _drop_maps(self, local: bool):
        if local:
            self._vertex_map = self._normal_map = None
        self._global_vertex_map = self._global_normal_map = None

    @rgb_image.setter
    def rgb_image(self, value):
        if value is not None:
            self._assert_shape(value, self._rgb_image_shape)
        self._rgb_image = value

    @depth_image.setter
    def depth_image(self, value):
        if value is not None:
            self._assert_shape(value, self._depth_image_shape)
        self._depth_image = value
        self._valid_depth_mask = None
        self._drop_maps(True)

    @intrinsics.setter
    def intrinsics(self, value):
        if value is not None:
            self._assert_shape(value, self._intrinsics_shape)
        self._intrinsics = value
        self._drop_maps(True)

    @poses.setter
    def poses(self, value):
        if value is not None:
            self._assert_shape(value, self._poses_shape)
        self._poses = value
        self._drop_maps(False)

    # ------------------------------------------------------------------ copies / moves
    def _copy_caches_to(self, other, fn):
        for k in self._INTERNAL_TENSORS:
            if k in ("_rgb_image", "_depth_image", "_intrinsics"):
                continue
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(other, k, fn(v))

    def clone(self):
        other = RGBDImages(self._rgb_image.clone(), self._depth_image.clone(), self._intrinsics.clone(),
                           channels_first=self.channels_first)
        self._copy_caches_to(other, lambda v: v.clone())
        return other

    def detach(self):
        other = self.clone()
        for k in self._INTERNAL_TENSORS:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(other, k, v.detach())
        return other

    def to(self, device: Union[torch.device, str], copy: bool = False):
        device = _resolve(device)
        if not copy and self.device == device:
            return self
        other = self.clone()
        other.device = device
        for k in self._INTERNAL_TENSORS:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(other, k, v.to(device))
        return other

    def cpu(self):
        return self.to(torch.device("cpu"))

    def cuda(self):
        return self.to(torch.device("cuda"))

    # ------------------------------------------------------------------ layout
    def to_channels_last(self, copy: bool = False):
        if not (copy or self.channels_first):
            return self
        return self.clone().to_channels_last_()

    def to_channels_first(self, copy: bool = False):
        if not copy and self.channels_first:
            return self
        return self.clone().to_channels_first_()

    def _permute_all(self, order, channels_first: bool):
        for k in ("_rgb_image", "_depth_image", "_vertex_map", "_global_vertex_map", "_normal_map", "_global_normal_map"):
            setattr(self, k, RGBDImages._permute_if_not_None(getattr(self, k), order))
        self._valid_depth_mask = None
        self._channels_first = channels_first
        self._rgb_image_shape = tuple(self._rgb_image.shape)
        self._depth_image_shape = tuple(self._depth_image.shape)
        return self

    def to_channels_last_(self):
        return self if not self.channels_first else self._permute_all((0, 1, 3, 4, 2), False)

    def to_channels_first_(self):
        return self if self.channels_first else self._permute_all((0, 1, 4, 2, 3), True)

    @staticmethod
    def _permute_if_not_None(tensor: Optional[torch.Tensor], ordering: tuple, contiguous: bool = True):
        if tensor is None:
            return None
        assert torch.is_tensor(tensor)
        out = tensor.permute(*ordering)
        return out.contiguous() if contiguous else out

    def plotly(self, index: int, include_depth: bool = True, as_figure: bool = True, ms_per_frame: int = 50):
        """`index`-th sequence as an animated plotly figure (play / stop buttons and a frame slider; RGB on top,
        depth below), or as the list of frame dicts `go.Figure(frames=...)` takes (reference :764-900).  Host side:
        one device-to-host copy of the images."""
        from plotly.subplots import make_subplots

        from .structutils import numpy_to_plotly_image

        if not isinstance(index, int):
            raise TypeError("Index should be int, but was {}.".format(type(index)))
        src = self.to_channels_last() if self.channels_first else self
        rgb = src.rgb_image[index]
        if (rgb.max() < 1.1).item():
            rgb = rgb * 255
        rgb = torch.clamp(rgb, min=0.0, max=255.0).detach().cpu().numpy().astype("uint8")
        rgb_images = [numpy_to_plotly_image(im, i) for i, im in enumerate(rgb)]
        if include_depth:
            depth = src.depth_image[index, ..., 0]
            scale = 10 ** torch.log10(255.0 / depth.detach().max()).floor().item()
            depth = (depth * scale).detach().cpu().numpy().astype("uint8")
            depth_images = [numpy_to_plotly_image(im, i, True, scale) for i, im in enumerate(depth)]
            frames = [{"name": i, "data": [a, b], "traces": [0, 1]} for i, (a, b) in enumerate(zip(rgb_images, depth_images))]
        else:
            frames = [{"data": [a], "name": i} for i, a in enumerate(rgb_images)]
        if not as_figure:
            return frames

        def animate(duration):
            return {"frame": {"duration": duration, "redraw": True}, "mode": "immediate", "fromcurrent": True,
                    "transition": {"duration": duration, "easing": "linear"}}

        slider = {"active": 0, "yanchor": "top", "xanchor": "left", "currentvalue": {"prefix": "Frame: "},
                  "pad": {"b": 10, "t": 60}, "len": 0.9, "x": 0.1, "y": 0,
                  "steps": [{"args": [[i], animate(0)], "label": i, "method": "animate"} for i in range(self._L)]}
        buttons = {"buttons": [{"args": [None, animate(ms_per_frame)], "label": "&#9654;", "method": "animate"},
                               {"args": [[None], animate(0)], "label": "&#9724;", "method": "animate"}],
                   "direction": "left", "pad": {"r": 10, "t": 70}, "showactive": False, "type": "buttons", "x": 0.1,
                   "xanchor": "right", "y": 0, "yanchor": "top"}
        if include_depth:
            fig = make_subplots(rows=2, cols=1, subplot_titles=("RGB", "Depth"), shared_xaxes=True, shared_yaxes=False,
                                vertical_spacing=0.1)
            fig.add_trace(frames[0]["data"][0], row=1, col=1)
            fig.add_trace(frames[0]["data"][1], row=2, col=1)
            fig.update_layout(scene=dict(aspectmode="data"))
            fig.update_layout(autosize=False, height=1080)
        else:
            fig = make_subplots(rows=1, cols=1, subplot_titles=("RGB",))
            fig.add_traces(frames[0]["data"][0])
        fig.update(frames=frames)
        fig.update_layout(updatemenus=[buttons], sliders=[slider])
        return fig
