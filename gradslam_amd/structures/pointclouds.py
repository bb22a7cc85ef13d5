"""Pointclouds: ragged batch of point clouds with per-point normals / colours / features.

Same public surface, attribute names, error contracts and caching behaviour as the reference
container (structures/pointclouds.py), because the hot-path functions and their tests reach into
it (`points_list`, `points_padded`, `_normals_list`, `has_features`, `append_points`, ...).  It
is host-side plumbing around HBM tensors: list <-> padded bookkeeping, views and concatenation.
The per-point arithmetic of the hot path lives in the HIP kernels (gradslam_amd/csrc).

Differences that are deliberate and invisible through the API:
* point counts are tracked as Python ints (`_counts`) so that no `.item()` device sync is needed to
  answer shape questions; `num_points_per_pointcloud` still returns a device tensor;
* the zero-padding check of the padded setters is one fused device reduction instead of a
  per-batch `.item()` loop (reference :1419-1427);
* the four attribute families share one implementation instead of four copies.
"""
from typing import List, Optional, Union

import torch

from ..geometry import projutils
from . import structutils

__all__ = ["Pointclouds"]

_ATTRS = ("points", "normals", "colors", "features")


def _resolve(device):
    return torch.Tensor().to(device).device


class Pointclouds(object):
    _INTERNAL_TENSORS = ["_points_padded", "_normals_padded", "_colors_padded", "_features_padded", "_nonpad_mask",
                         "_num_points_per_pointcloud"]

    def __init__(self, points=None, normals=None, colors=None, features=None, device=None):
        super().__init__()
        if not (points is None or isinstance(points, list) or torch.is_tensor(points)):
            raise TypeError("Expected points to be of type list or tensor or None; got %r" % type(points))
        for name, val in (("normals", normals), ("colors", colors), ("features", features)):
            if not (val is None or isinstance(val, type(points))):
                raise TypeError("Expected %s to be of same type as points (%r); got %r" % (name, type(points), type(val)))
        if points is not None and len(points) == 0:
            raise ValueError("len(points) (= 0) should be > 0")

        for a in _ATTRS:
            setattr(self, "_%s_list" % a, None)
            setattr(self, "_%s_padded" % a, None)
            setattr(self, "_has_%s" % a, None)
        self._nonpad_mask = None
        self._num_points_per_pointcloud = None
        self._counts: List[int] = [0]
        self.equisized = False

        if isinstance(points, list):
            shapes = [p.shape for p in points]
            if any(p.ndim != 2 for p in points):
                raise ValueError("ndim of all tensors in points list should be 2")
            if any(s[-1] != 3 for s in shapes):
                raise ValueError("last dim of all tensors in points should have shape 3 (X, Y, Z)")
            self.device = _resolve(device) if device is not None else points[0].device
            counts = [int(s[0]) for s in shapes]
            if not (normals is None or [n.shape for n in normals] == shapes):
                raise ValueError("normals tensors should have same shape as points tensors, but didn't")
            if not (colors is None or [c.shape for c in colors] == shapes):
                raise ValueError("colors tensors should have same shape as points tensors, but didn't")
            if not (features is None or all(f.ndim == 2 for f in features)):
                raise ValueError("ndim of all tensors in features list should be 2")
            if not (features is None or [len(f) for f in features] == counts):
                raise ValueError("number of features per pointcloud has to be equal to number of points")
            if not (features is None or len(set(f.shape[-1] for f in features)) == 1):
                raise ValueError("number of features per pointcloud has to be the same")
            to_dev = lambda xs: None if xs is None else [x.to(self.device) for x in xs]
            self._points_list, self._normals_list = to_dev(points), to_dev(normals)
            self._colors_list, self._features_list = to_dev(colors), to_dev(features)
            self._B = len(points)
            self._set_counts(counts)

        elif torch.is_tensor(points):
            self.device = _resolve(device) if device is not None else points.device
            if points.ndim != 3:
                raise ValueError("points should have ndim=3, but had ndim={}".format(points.ndim))
            if points.shape[-1] != 3:
                raise ValueError("last dim of points should have shape 3 (X, Y, Z) but had shape %r" % (points.shape[-1]))
            if points.shape[0] == 0:
                raise ValueError("Batch size of 0 not supported yet. Got input points shape {}.".format(points.shape))
            if not (normals is None or normals.shape == points.shape):
                raise ValueError("normals tensor should have same shape as points tensor, but didn't: %r != %r"
                                 % (normals.shape, points.shape))
            if not (colors is None or colors.shape == points.shape):
                raise ValueError("colors tensor should have same shape as points tensor, but didn't: %r != %r"
                                 % (colors.shape, points.shape))
            if not (features is None or features.ndim == 3):
                raise ValueError("features should have ndim=3, but had ndim={}".format(features.ndim))
            if not (features is None or features.shape[:-1] == points.shape[:-1]):
                raise ValueError("first 2 dims of features tensor and points tensor should have same shape, but didn't: "
                                 "%r != %r" % (features.shape[:-1], points.shape[:-1]))
            mv = lambda x: None if x is None else x.to(self.device)
            self._points_padded, self._normals_padded = mv(points), mv(normals)
            self._colors_padded, self._features_padded = mv(colors), mv(features)
            self._B = points.shape[0]
            self._set_counts([int(points.shape[1])] * self._B)

        else:  # empty
            self.device = _resolve(device) if device is not None else torch.device("cpu")
            self._B = 0
            self._N = 0
            self._counts = [0]
            self.equisized = None

    # ------------------------------------------------------------------ bookkeeping
    def _set_counts(self, counts: List[int]):
        self._counts = [int(c) for c in counts]
        self._N = max(self._counts)
        self.equisized = len(set(self._counts)) == 1
        self._num_points_per_pointcloud = None  # device mirror, built on demand
        self._nonpad_mask = None

    def __len__(self):
        return self._B

    @property
    def num_points_per_pointcloud(self):
        if self._num_points_per_pointcloud is None:
            self._num_points_per_pointcloud = torch.tensor(self._counts, device=self.device)
        return self._num_points_per_pointcloud

    def _counts_i32(self) -> torch.Tensor:
        """int32 device copy of the per-batch counts, the form the kernels read (cached)."""
        cached = getattr(self, "_counts_i32_cache", None)
        if cached is None or cached[0] != self._counts:
            if len(self._counts) == 1:  # filled by a kernel: no pageable host->device copy on the frame path
                dev_counts = torch.full((1,), int(self._counts[0]), dtype=torch.int32, device=self.device)
            else:
                dev_counts = torch.tensor(self._counts, dtype=torch.int32, device=self.device)
            cached = (list(self._counts), dev_counts)
            self._counts_i32_cache = cached
        return cached[1]

    @property
    def nonpad_mask(self):
        if self._nonpad_mask is None and self.has_points:
            ar = torch.arange(self._N, device=self.device).unsqueeze(0)
            self._nonpad_mask = ar < self.num_points_per_pointcloud.unsqueeze(1)
        return self._nonpad_mask

    def _has(self, a: str) -> bool:
        flag = getattr(self, "_has_%s" % a)
        if flag is None:
            flag = getattr(self, "_%s_list" % a) is not None or getattr(self, "_%s_padded" % a) is not None
            setattr(self, "_has_%s" % a, flag)
        return flag

    has_points = property(lambda self: self._has("points"))
    has_normals = property(lambda self: self._has("normals"))
    has_colors = property(lambda self: self._has("colors"))
    has_features = property(lambda self: self._has("features"))

    @property
    def num_features(self):
        if not self.has_features:
            return 0
        if self._features_padded is not None:
            return self._features_padded.shape[-1]
        return self._features_list[0].shape[-1]

    # ------------------------------------------------------------------ list / padded views
    def _get_list(self, a: str):
        cur = getattr(self, "_%s_list" % a)
        pad = getattr(self, "_%s_padded" % a)
        if cur is None and pad is not None:
            cur = [pad[b, : self._counts[b]] for b in range(self._B)]
            setattr(self, "_%s_list" % a, cur)
        return cur

    def _compute_padded(self, refresh: bool = False):
        if not self.has_points or not (refresh or self._points_padded is None):
            return
        for a in _ATTRS:
            lst = getattr(self, "_%s_list" % a)
            if lst is None:
                setattr(self, "_%s_padded" % a, None)
                continue
            width = 3 if a != "features" else self.num_features
            if self._B == 1:
                # one cloud: the padded form IS the list item (a view, no copy of a 10^5..10^6-point map
                # per frame); nothing in this package writes into either form in place
                setattr(self, "_%s_padded" % a, lst[0].unsqueeze(0))
                continue
            setattr(self, "_%s_padded" % a,
                    structutils.list_to_padded(lst, (self._N, width), pad_value=0.0, equisized=self.equisized))

    def _get_padded(self, a: str):
        self._compute_padded()
        return getattr(self, "_%s_padded" % a)

    def _set_list(self, a: str, value):
        self._assert_set_list(value, first_dim_only=(a == "features"))
        setattr(self, "_%s_list" % a, [v.clone().to(self.device) for v in value])
        # NB the reference only drops the padded cache for `points` (its other three setters assign
        # to a misspelt attribute, structures/pointclouds.py:850,864,878); the stale cache is
        # observable, so it is kept.
        if a == "points":
            self._points_padded = None

    def _set_padded(self, a: str, value):
        self._assert_set_padded(value, first_2_dims_only=(a == "features"))
        setattr(self, "_%s_padded" % a, value.clone().to(self.device))
        setattr(self, "_%s_list" % a, None)

    points_list = property(lambda s: s._get_list("points"), lambda s, v: s._set_list("points", v))
    normals_list = property(lambda s: s._get_list("normals"), lambda s, v: s._set_list("normals", v))
    colors_list = property(lambda s: s._get_list("colors"), lambda s, v: s._set_list("colors", v))
    features_list = property(lambda s: s._get_list("features"), lambda s, v: s._set_list("features", v))
    points_padded = property(lambda s: s._get_padded("points"), lambda s, v: s._set_padded("points", v))
    normals_padded = property(lambda s: s._get_padded("normals"), lambda s, v: s._set_padded("normals", v))
    colors_padded = property(lambda s: s._get_padded("colors"), lambda s, v: s._set_padded("colors", v))
    features_padded = property(lambda s: s._get_padded("features"), lambda s, v: s._set_padded("features", v))

    def _assert_set_padded(self, value, first_2_dims_only: bool = False):
        if not isinstance(value, torch.Tensor):
            raise TypeError("value must be torch.Tensor. Got {}".format(type(value)))
        if not self.has_points:
            raise ValueError("cannot set padded representation for an empty pointclouds object")
        if self.device != torch.device(value.device):
            raise ValueError("value must have the same device as pointclouds object: {} != {}".format(
                value.device, torch.device(self.device)))
        if value.ndim != 3:
            raise ValueError("value.ndim should be 3. Got {}".format(value.ndim))
        ref = self.points_padded.shape
        if first_2_dims_only and ref[:2] != value.shape[:2]:
            raise ValueError("first 2 dims of value tensor and points tensor should have same shape, but didn't: "
                             "{} != {}.".format(value.shape[:2], ref[:2]))
        if (not first_2_dims_only) and ref != value.shape:
            raise ValueError("value tensor and points tensor should have same shape, but didn't: {} != {}.".format(
                value.shape, ref))
        if not self.equisized or self._counts[0] != self._N:
            pad = ~self.nonpad_mask
            if bool((value.detach().ne(0) & pad.unsqueeze(-1)).any()):
                raise ValueError("value must have zeros wherever pointclouds.points_padded has zero padding.")

    def _assert_set_list(self, value, first_dim_only: bool = False):
        if not isinstance(value, list):
            raise TypeError("value must be list of torch.Tensors. Got {}".format(type(value)))
        if not self.has_points:
            raise ValueError("cannot set list representation for an empty pointclouds object")
        if len(self) != len(value):
            raise ValueError("value must have same length as pointclouds.points_list. Got {} != {}.".format(
                len(value), len(self)))
        if any(v.ndim != 2 for v in value):
            raise ValueError("ndim of all tensors in value list should be 2")
        mine = self.points_list
        if first_dim_only and any(mine[b].shape[:1] != value[b].shape[:1] for b in range(len(self))):
            raise ValueError("shape of first 2 dims of tensors in value and pointclouds.points_list must match")
        if (not first_dim_only) and any(mine[b].shape != value[b].shape for b in range(len(self))):
            raise ValueError("shape of tensors in value and pointclouds.points_list must match")

    def _adopt_padded(self, points, normals, colors, features):
        """Install freshly computed padded attributes (same shapes, padding already zero) without the
        clone + zero-padding re-check of the public setters: the merge kernel wrote them."""
        self._points_padded, self._normals_padded = points, normals
        self._colors_padded, self._features_padded = colors, features
        self._points_list = self._normals_list = self._colors_list = self._features_list = None

    def _adopt_rows(self, arrays, counts: List[int]):
        """Become the clouds held in arena arrays: arrays = (points, normals, colors, features-or-None), each
        (B, cap, C) with rows beyond counts[b] zero.  The list items are views of the arrays (no copy)."""
        self._B = int(arrays[0].shape[0])
        self.device = arrays[0].device
        for a, arr in zip(_ATTRS, arrays):
            setattr(self, "_%s_list" % a, None if arr is None else [arr[b, : counts[b]] for b in range(self._B)])
            setattr(self, "_%s_padded" % a, None)
            setattr(self, "_has_%s" % a, arr is not None)
        self._set_counts(counts)
        return self

    # ------------------------------------------------------------------ indexing
    def __getitem__(self, index):
        if not self.has_points:
            raise IndexError("Cannot index empty pointclouds object")
        if isinstance(index, int):
            pick = lambda xs: [xs[index]]
        elif isinstance(index, slice):
            pick = lambda xs: xs[index]
        elif isinstance(index, list):
            pick = lambda xs: [xs[i] for i in index]
        elif isinstance(index, torch.Tensor):
            if index.dim() != 1 or index.dtype.is_floating_point:
                raise IndexError(index)
            if index.dtype == torch.bool:
                index = index.nonzero()
                index = index.squeeze(1) if index.numel() > 0 else index
            ids = index.tolist()
            pick = lambda xs: [xs[i] for i in ids]
        else:
            raise IndexError(index)
        sel = {a: (pick(self._get_list(a)) if self._has(a) else None) for a in _ATTRS}
        return Pointclouds(points=sel["points"], normals=sel["normals"], colors=sel["colors"], features=sel["features"])

    # ------------------------------------------------------------------ arithmetic
    def __add__(self, other):
        try:
            return self.clone().offset_(other)
        except TypeError:
            raise NotImplementedError("Pointclouds + {} currently not implemented.".format(type(other)))

    def __sub__(self, other):
        try:
            return self.clone().offset_(other * -1)
        except TypeError:
            raise NotImplementedError("Pointclouds - {} currently not implemented.".format(type(other)))

    def __mul__(self, other):
        try:
            return self.clone().scale_(other)
        except TypeError:
            raise NotImplementedError("Pointclouds * {} currently not implemented.".format(type(other)))

    def __truediv__(self, other):
        try:
            return self.__mul__(1.0 / other)
        except TypeError:
            raise NotImplementedError("Pointclouds / {} currently not implemented.".format(type(other)))

    def __matmul__(self, other):
        if not torch.is_tensor(other):
            raise NotImplementedError("Pointclouds @ {} currently not implemented.".format(type(other)))
        if not ((other.ndim == 2 or other.ndim == 3) and (other.shape[-2:] == (3, 3) or other.shape[-2:] == (4, 4))):
            msg = "Unsupported shape for Pointclouds @ operand: {}\n".format(other.shape)
            msg += "Use tensor of shape (3, 3) or (B, 3, 3) for rotations, or (4, 4) or (B, 4, 4) for transformations"
            raise ValueError(msg)
        if other.shape[-2:] == (3, 3):
            return self.clone().rotate_(other, pre_multiplication=False)
        return self.clone().transform_(other, pre_multiplication=False)

    def rotate(self, rmat, *, pre_multiplication=True):
        return self.clone().rotate_(rmat, pre_multiplication=pre_multiplication)

    def transform(self, transform, *, pre_multiplication=True):
        return self.clone().transform_(transform, pre_multiplication=pre_multiplication)

    def pinhole_projection(self, intrinsics):
        return self.clone().pinhole_projection_(intrinsics)

    def _padmask_f(self):
        return self.nonpad_mask.to(self.points_padded.dtype).unsqueeze(-1)

    def offset_(self, offset: Union[torch.Tensor, float, int]):
        if not (torch.is_tensor(offset) or isinstance(offset, (float, int))):
            raise TypeError("Operand should be tensor, float or int but was %r instead" % type(offset))
        if not self.has_points:
            return self
        self._points_padded = self.points_padded + (offset * self._padmask_f())
        self._points_list = None
        return self

    def scale_(self, scale: Union[torch.Tensor, float, int]):
        if not (torch.is_tensor(scale) or isinstance(scale, (float, int))):
            raise TypeError("Operand should be tensor, float or int but was %r instead" % type(scale))
        if not self.has_points:
            return self
        self._points_padded = self.points_padded * scale * self._padmask_f()
        self._points_list = None
        return self

    def rotate_(self, rmat: torch.Tensor, *, pre_multiplication=True):
        if not torch.is_tensor(rmat):
            raise TypeError("Rotation matrix should be tensor, but was %r instead" % type(rmat))
        if not ((rmat.ndim == 2 or rmat.ndim == 3) and rmat.shape[-2:] == (3, 3)):
            raise ValueError("Rotation matrix should be of shape (3, 3) or (B, 3, 3), but was {} instead.".format(rmat.shape))
        if rmat.ndim == 3 and rmat.shape[0] != self._B:
            raise ValueError("Rotation matrix batch size ({}) != Pointclouds batch size ({})".format(rmat.shape[0], self._B))
        if not self.has_points:
            return self
        if pre_multiplication:
            rmat = rmat.transpose(-1, -2)
        spec = "bij,jk->bik" if rmat.ndim == 2 else "bij,bjk->bik"
        self._points_padded = torch.einsum(spec, self.points_padded, rmat)
        self._normals_padded = None if self.normals_padded is None else torch.einsum(spec, self.normals_padded, rmat)
        self._points_list = None
        self._normals_list = None
        return self

    def transform_(self, transform: torch.Tensor, *, pre_multiplication=True):
        if not torch.is_tensor(transform):
            raise TypeError("transform should be tensor, but was %r instead" % type(transform))
        if not ((transform.ndim == 2 or transform.ndim == 3) and transform.shape[-2:] == (4, 4)):
            raise ValueError("transform should be of shape (4, 4) or (B, 4, 4), but was {} instead.".format(transform.shape))
        if transform.ndim == 3 and transform.shape[0] != self._B:
            raise ValueError("transform batch size ({}) != Pointclouds batch size ({})".format(transform.shape[0], self._B))
        if not self.has_points:
            return self
        rmat, tvec = transform[..., :3, :3], transform[..., :3, 3]
        while tvec.ndim < self.points_padded.ndim:
            tvec = tvec.unsqueeze(-2)
        return self.rotate_(rmat, pre_multiplication=pre_multiplication).offset_(tvec)

    def pinhole_projection_(self, intrinsics: torch.Tensor):
        if not torch.is_tensor(intrinsics):
            raise TypeError("intrinsics should be tensor, but was {} instead".format(type(intrinsics)))
        if not ((intrinsics.ndim == 2 or intrinsics.ndim == 3) and intrinsics.shape[-2:] == (4, 4)):
            raise ValueError("intrinsics should be of shape (4, 4) or (B, 4, 4), but was {} instead.".format(intrinsics.shape))
        if not self.has_points:
            return self
        uv = projutils.project_points(self.points_padded, intrinsics)
        self._points_padded = projutils.homogenize_points(uv) * self.nonpad_mask.to(uv.dtype).unsqueeze(-1)
        self._points_list = None
        return self

    # ------------------------------------------------------------------ copies
    def clone(self):
        if not self.has_points:
            return Pointclouds(device=self.device)
        if self._points_list is not None:
            cp = lambda xs: None if xs is None else [x.clone() for x in xs]
            new = {a: cp(getattr(self, "_%s_list" % a) if a != "points" else self.points_list) for a in _ATTRS}
        else:
            cp = lambda x: None if x is None else x.clone()
            new = {a: cp(getattr(self, "_%s_padded" % a)) for a in _ATTRS}
        other = Pointclouds(points=new["points"], normals=new["normals"], colors=new["colors"], features=new["features"])
        other._set_counts(self._counts)
        for k in self._INTERNAL_TENSORS:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(other, k, v.clone())
        return other

    def detach(self):
        other = self.clone()
        for a in _ATTRS:
            lst = getattr(other, "_%s_list" % a)
            if lst is not None:
                setattr(other, "_%s_list" % a, [x.detach() for x in lst])
        for k in self._INTERNAL_TENSORS:
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(other, k, v.detach())
        return other

    def to(self, device, copy: bool = False):
        if not copy and self.device == device:
            return self
        other = self.clone()
        if self.device != device:
            other.device = _resolve(device)
            for a in _ATTRS:
                lst = getattr(other, "_%s_list" % a)
                if lst is not None:
                    setattr(other, "_%s_list" % a, [x.to(device) for x in lst])
            for k in self._INTERNAL_TENSORS:
                v = getattr(self, k)
                if torch.is_tensor(v):
                    setattr(other, k, v.to(device))
        return other

    def cpu(self):
        return self.to(torch.device("cpu"))

    def cuda(self):
        return self.to(torch.device("cuda"))

    # ------------------------------------------------------------------ append
    def append_points(self, pointclouds: "Pointclouds"):
        if not isinstance(pointclouds, type(self)):
            raise TypeError("Append object must be of type gradslam.Pointclouds, but was of type {}.".format(type(pointclouds)))
        if not (pointclouds.device == self.device):
            raise ValueError("Device of pointclouds to append and to be appended must match: ({0} != {1})".format(
                pointclouds.device, self.device))
        if not pointclouds.has_points:
            return self
        if not self.has_points:
            for a in _ATTRS:
                if pointclouds._has(a):
                    setattr(self, "_%s_list" % a, [x.clone().to(self.device) for x in pointclouds._get_list(a)])
                setattr(self, "_has_%s" % a, getattr(pointclouds, "_has_%s" % a))
            self._B = pointclouds._B
            self._set_counts(pointclouds._counts)
            for k in self._INTERNAL_TENSORS:
                v = getattr(pointclouds, k)
                if torch.is_tensor(v):
                    setattr(self, k, v.clone())
            return self
        if not (len(pointclouds) == len(self)):
            raise ValueError("Batch size of pointclouds to append and to be appended must match: ({0} != {1})".format(
                len(pointclouds), len(self)))
        for a in ("normals", "colors", "features"):
            if self._has(a) != pointclouds._has(a):
                raise ValueError("pointclouds to append and to be appended must either both have or not have {2}: "
                                 "({0} != {1})".format(pointclouds._has(a), self._has(a), a))
        if self.has_features and self.num_features != pointclouds.num_features:
            raise ValueError("pointclouds to append and to be appended must have the same number of features: "
                             "({0} != {1})".format(pointclouds.num_features, self.num_features))
        for a in _ATTRS:
            if not self._has(a):
                continue
            mine, theirs = self._get_list(a), pointclouds._get_list(a)
            setattr(self, "_%s_list" % a, [torch.cat([mine[b], theirs[b]], 0) for b in range(self._B)])
            setattr(self, "_%s_padded" % a, None)
        self._set_counts([x + y for x, y in zip(self._counts, pointclouds._counts)])
        return self

    # ------------------------------------------------------------------ export (out of scope: viewers)
    def open3d(self, index: int, include_colors: bool = True, max_num_points: Optional[int] = None,
               include_normals: bool = False):
        """`index`-th cloud as an `open3d.geometry.PointCloud` (reference :1239-1294): a device-to-host copy;
        colours above 1.1 are taken to be 0..255 and normalised.  Needs the `open3d` package (not in this image)."""
        import open3d as o3d

        if not isinstance(index, int):
            raise TypeError("Index should be int, but was {}.".format(type(index)))
        n = self.points_list[index].shape[0]
        keep = None
        if max_num_points is not None and max_num_points < n:
            keep = torch.randperm(n)[:max_num_points].to(self.device)
        take = lambda x: (x if keep is None else x[keep]).detach().cpu().numpy()
        pcd = o3d.geometry.PointCloud()
        pcd.points = o3d.utility.Vector3dVector(take(self.points_list[index]))
        if self.has_colors and include_colors:
            colors = self.colors_list[index]
            if (colors.max() > 1.1).item():
                colors = colors / 255
            pcd.colors = o3d.utility.Vector3dVector(take(torch.clamp(colors, min=0.0, max=1.0)))
        if self.has_normals and include_normals:
            pcd.normals = o3d.utility.Vector3dVector(take(self.normals_list[index]))
        return pcd

    def plotly(self, index: int, include_colors: bool = True, max_num_points: Optional[int] = 200000, as_figure: bool = True,
               point_size: int = 2):
        """`index`-th cloud as a `plotly.graph_objects.Figure` (or `Scatter3d` if not `as_figure`) for viewing
        (reference :1296-1383): a device-to-host copy of at most `max_num_points` randomly chosen points; colours
        in [0, 1.1) are taken to be normalised and scaled to 0..255."""
        import plotly.graph_objects as go

        if not isinstance(index, int):
            raise TypeError("Index should be int, but was {}.".format(type(index)))
        points = self.points_list[index]
        n = points.shape[0]
        subsample = max_num_points is not None and max_num_points < n
        if subsample:
            keep = torch.randperm(n)[:max_num_points].to(points.device)
            points = points[keep]
        xyz = points.detach().cpu().numpy()
        marker = {"size": point_size}
        if self.has_colors and include_colors:
            colors = self.colors_list[index]
            if subsample:
                colors = colors[keep]
            if (colors.max() < 1.1).item():
                colors = colors * 255
            marker["color"] = torch.clamp(colors, min=0.0, max=255.0).detach().cpu().numpy().astype("uint8")
        scatter = go.Scatter3d(x=xyz[..., 0], y=xyz[..., 1], z=xyz[..., 2], mode="markers", marker=marker)
        if not as_figure:
            return scatter
        hidden = dict(showticklabels=False, showgrid=False, zeroline=False, visible=False)
        fig = go.Figure(data=[scatter])
        fig.update_layout(showlegend=False, scene=dict(xaxis=hidden, yaxis=hidden, zaxis=hidden))
        return fig
