"""What the reference's RGB-D sequence loaders (datasets/tum.py, icl.py) have in common: the sequence slicing
arguments, the `__getitem__` tuple, the per-frame conversion -- on the host like the reference, or on the
device from the raw integer frames (`load_rgbdimages`)."""
from typing import Optional, Union

import numpy as np
import torch
from torch.utils import data

from ..geometry.geometryutils import relative_transformation
from . import datautils


def imread(path: str) -> np.ndarray:
    """uint16 (H, W) for 16-bit depth PNGs, uint8 (H, W, 3) for colour (PIL; the reference uses imageio)."""
    from PIL import Image

    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            return np.asarray(im).astype(np.uint16)
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def resize_nearest(a: np.ndarray, h: int, w: int) -> np.ndarray:
    if a.shape[:2] == (h, w):
        return a
    ys = np.minimum(np.floor(np.arange(h) * (a.shape[0] / h)).astype(np.int64), a.shape[0] - 1)
    xs = np.minimum(np.floor(np.arange(w) * (a.shape[1] / w)).astype(np.int64), a.shape[1] - 1)
    return a[ys][:, xs]


def resize_bilinear(a: np.ndarray, h: int, w: int) -> np.ndarray:
    """float64 (H, W, C) -> (h, w, C); pixel centres at +0.5, edges clamped (the gs_frames_from_raw rule)."""
    if a.shape[:2] == (h, w):
        return a

    def taps(n_out, n_in):
        f = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(f).astype(np.int64)
        f = f - i0
        f[i0 < 0] = 0.0
        i0 = np.maximum(i0, 0)
        f[i0 >= n_in - 1] = 0.0
        i0 = np.minimum(i0, n_in - 1)
        return i0, np.minimum(i0 + 1, n_in - 1), f

    y0, y1, fy = taps(h, a.shape[0])
    x0, x1, fx = taps(w, a.shape[1])
    fx, fy = fx[None, :, None], fy[:, None, None]
    top = a[y0][:, x0] * (1.0 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1.0 - fx) + a[y1][:, x1] * fx
    return top * (1.0 - fy) + bot * fy


class SequenceDataset(data.Dataset):
    """Subclasses fill `colorfiles`, `depthfiles`, `framenames` (one entry per extracted sequence), set
    `intrinsics` (1, 4, 4), `scaling_factor`, and implement `_sequence_poses(idx) -> list of (4, 4) float
    arrays`."""

    orthogonal_rotations_default = True  # TUM calls relative_transformation with its default, ICL with False

    def _init_common(self, seqlen, dilation, stride, start, end, height, width, channels_first, normalize_color, return_depth,
                     return_intrinsics, return_pose, return_transform, return_names):
        self.height, self.width = height, width
        self.height_downsample_ratio = float(height) / 480
        self.width_downsample_ratio = float(width) / 640
        self.channels_first = channels_first
        self.normalize_color = normalize_color
        self.return_depth, self.return_intrinsics = return_depth, return_intrinsics
        self.return_pose, self.return_transform, self.return_names = return_pose, return_transform, return_names
        self.load_poses = self.return_pose or self.return_transform
        if not isinstance(seqlen, int):
            raise TypeError('"seqlen" must be int. Got {0}.'.format(type(seqlen)))
        if not (isinstance(stride, int) or stride is None):
            raise TypeError('"stride" must be int or None. Got {0}.'.format(type(stride)))
        if not (isinstance(dilation, int) or dilation is None):
            raise TypeError("dilation must be int or None. Got {0}.".format(type(dilation)))
        dilation = dilation if dilation is not None else 0
        stride = stride if stride is not None else seqlen * (dilation + 1)
        self.seqlen, self.stride, self.dilation = seqlen, stride, dilation
        if seqlen < 0:
            raise ValueError('"seqlen" must be positive. Got {0}.'.format(seqlen))
        if dilation < 0:
            raise ValueError('"dilation" must be positive. Got {0}.'.format(dilation))
        if stride < 0:
            raise ValueError('"stride" must be positive. Got {0}.'.format(stride))
        if not (isinstance(start, int) or start is None):
            raise TypeError('"start" must be int or None. Got {0}.'.format(type(start)))
        if not (isinstance(end, int) or end is None):
            raise TypeError('"end" must be int or None. Got {0}.'.format(type(end)))
        start = start if start is not None else 0
        self.start, self.end = start, end
        if start < 0:
            raise ValueError('"start" must be None or positive. Got {0}.'.format(stride))
        if not (end is None or end > start):
            raise ValueError('"end" ({0}) must be None or greater than start ({1})'.format(end, start))

    def _windows(self, n_frames: int):
        """Index arrays of the sequences cut from a trajectory of n_frames frames."""
        idx = np.arange(self.seqlen) * (self.dilation + 1)
        for first in range(0, n_frames, self.stride):
            if first + idx[-1] >= n_frames:
                break
            yield first + idx

    def __len__(self):
        return self.num_sequences

    # ------------------------------------------------------------------ host path (reference semantics)
    def __getitem__(self, idx: int):
        color_seq, depth_seq = [], []
        for i in range(self.seqlen):
            color_seq.append(torch.from_numpy(self._preprocess_color(imread(self.colorfiles[idx][i]).astype(float))))
            if self.return_depth:
                depth_seq.append(torch.from_numpy(self._preprocess_depth(imread(self.depthfiles[idx][i]).astype(np.int64))))
        output = [torch.stack(color_seq, 0).float()]
        if self.return_depth:
            output.append(torch.stack(depth_seq, 0).float())
        if self.return_intrinsics:
            output.append(self.intrinsics)
        poses = self._sequence_poses(idx) if self.load_poses else None
        if self.return_pose:
            output.append(self._preprocess_poses(torch.stack([torch.from_numpy(p) for p in poses], 0).float()))
        if self.return_transform:
            output.append(torch.stack([torch.from_numpy(np.asarray(x)).float() for x in datautils.poses_to_transforms(poses)], 0))
        if self.return_names:
            output.append(self.framenames[idx])
        return tuple(output) + self._extra_outputs(idx)

    def _extra_outputs(self, idx: int) -> tuple:
        return ()

    def _preprocess_color(self, color: np.ndarray):
        color = resize_bilinear(color, self.height, self.width)
        if self.normalize_color:
            color = datautils.normalize_image(color)
        if self.channels_first:
            color = datautils.channels_first(color)
        return color

    def _preprocess_depth(self, depth: np.ndarray):
        depth = np.expand_dims(resize_nearest(depth.astype(float), self.height, self.width), -1)
        if self.channels_first:
            depth = datautils.channels_first(depth)
        return depth / self.scaling_factor

    def _preprocess_poses(self, poses: torch.Tensor):
        """Poses relative to the first frame of the sequence (first one = identity)."""
        first = poses[0].unsqueeze(0).repeat(poses.shape[0], 1, 1)
        if self.orthogonal_rotations_default:
            return relative_transformation(first, poses)
        return relative_transformation(first, poses, orthogonal_rotations=False)

    # ------------------------------------------------------------------ device path
    def load_rgbdimages(self, idx: int, device: Union[str, torch.device] = "cuda:0"):
        """Sequence `idx` as an RGBDImages (1, L, H, W, C) on `device`: raw frames uploaded as uint8 / uint16 and
        converted by the HIP kernel (scale, resize, normalise); poses relative to the first frame."""
        from .. import ops
        from ..structures.rgbdimages import RGBDImages

        rgb_raw = torch.from_numpy(np.stack([imread(p) for p in self.colorfiles[idx]])).to(device)
        depth_raw = torch.from_numpy(np.stack([imread(p) for p in self.depthfiles[idx]]).view(np.int16)).to(device)
        depth, rgb = ops.frames_from_raw(depth_raw, rgb_raw, self.height, self.width, self.scaling_factor, self.normalize_color)
        poses = None
        if self.load_poses:
            poses = self._preprocess_poses(torch.stack([torch.from_numpy(np.asarray(p)) for p in self._sequence_poses(idx)],
                                                       0).float()).unsqueeze(0).to(device)
        frames = RGBDImages(rgb.unsqueeze(0), depth.unsqueeze(0), self.intrinsics.unsqueeze(0).to(device), poses)
        return frames.to_channels_first() if self.channels_first else frames
