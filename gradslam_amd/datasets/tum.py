"""TUM RGB-D sequences (reference datasets/tum.py:17-569): same constructor, same `__getitem__` tuple
(colour, depth, intrinsics, poses, transforms, names, timestamps; each optional), same sequence slicing
(seqlen / dilation / stride / start / end), TUM intrinsics fx=fy=525, cx=319.5, cy=239.5 scaled to the requested
size, depth = png / 5000.

Decoding: PIL on the host (the reference uses imageio + cv2, absent here).  Two ways out:
* `dataset[i]` -- CPU float tensors like the reference (resize on the host: nearest for depth, bilinear for
  colour; same-size frames are exact, resized colour is parity-unpinned against cv2.INTER_LINEAR);
* `dataset.load_rgbdimages(i, device)` -- the MI355X path: the raw uint8 / uint16 frames are uploaded (5 B per
  pixel instead of 16) and converted by `gs_frames_from_raw` on the device, returning an RGBDImages."""
import os
from typing import Optional, Union

import numpy as np
import torch
from torch.utils import data

from ..geometry.geometryutils import relative_transformation
from . import datautils, tumutils

__all__ = ["TUM"]

_DIRMSG = ("TUM folder should look something like:\n\n| ├── basedir\n| │   ├── rgbd_dataset_freiburgX_NAME\n"
           "| │   │   ├── depth/\n| │   │   ├── rgb/\n| │   │   ├── accelerometer.txt\n| │   │   └── depth.txt\n"
           "| │   │   └── groundtruth.txt\n| │   │   └── rgb.txt\n| │   ├── ...")


def _imread(path: str) -> np.ndarray:
    from PIL import Image

    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            return np.asarray(im).astype(np.uint16)
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def _resize_nearest(a: np.ndarray, h: int, w: int) -> np.ndarray:
    if a.shape[:2] == (h, w):
        return a
    ys = np.minimum(np.floor(np.arange(h) * (a.shape[0] / h)).astype(np.int64), a.shape[0] - 1)
    xs = np.minimum(np.floor(np.arange(w) * (a.shape[1] / w)).astype(np.int64), a.shape[1] - 1)
    return a[ys][:, xs]


def _resize_bilinear(a: np.ndarray, h: int, w: int) -> np.ndarray:
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


class TUM(data.Dataset):
    def __init__(self, basedir: str, sequences: Union[tuple, str, None] = None, seqlen: int = 4,
                 dilation: Optional[int] = None, stride: Optional[int] = None, start: Optional[int] = None,
                 end: Optional[int] = None, height: int = 480, width: int = 640, channels_first: bool = False,
                 normalize_color: bool = False, *, return_depth: bool = True, return_intrinsics: bool = True,
                 return_pose: bool = True, return_transform: bool = True, return_names: bool = True,
                 return_timestamps: bool = True):
        super().__init__()
        basedir = os.path.normpath(basedir)
        self.height, self.width = height, width
        self.height_downsample_ratio = float(height) / 480
        self.width_downsample_ratio = float(width) / 640
        self.channels_first = channels_first
        self.normalize_color = normalize_color
        self.return_depth, self.return_intrinsics = return_depth, return_intrinsics
        self.return_pose, self.return_transform = return_pose, return_transform
        self.return_names, self.return_timestamps = return_names, return_timestamps
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

        if isinstance(sequences, str):
            if not os.path.isfile(sequences):
                raise ValueError("incorrect filename: {} doesn't exist".format(sequences))
            with open(sequences, "r") as f:
                sequences = tuple(f.read().split("\n"))
        elif not (sequences is None or isinstance(sequences, tuple)):
            raise TypeError('"sequences" should either be path to .txt file or tuple of sequence names or None,  but was '
                            "of type {0} instead".format(type(sequences)))
        if isinstance(sequences, tuple) and len(sequences) == 0:
            raise ValueError('"sequences" must have atleast one element. Got len(sequences)=0')

        sequence_paths = []
        for item in sorted(os.listdir(basedir)):
            if not os.path.isdir(os.path.join(basedir, item)):
                continue
            split = item.split("_")
            if len(split) < 4 or split[0] != "rgbd" or split[1] != "dataset" or split[2][:-1] != "freiburg":
                raise ValueError('Incorrect folder names in "basedir" ({0}). Folder names of extracted .tgz files from TUM '
                                 'should follow the following naming convention: "rgbd_dataset_freiburgX_NAME". Got "{1}".'
                                 .format(basedir, item))
            if sequences is None or item in sequences:
                sequence_paths.append(os.path.join(basedir, item))
        if len(sequence_paths) == 0:
            raise ValueError('Incorrect folder structure in basedir ("{0}"). '.format(basedir) + _DIRMSG)
        if sequences is not None and len(sequence_paths) != len(sequences):
            raise ValueError('"sequences" contains sequences not available in basedir:\n"sequences" contains: '
                             + ", ".join(sequences) + '\n"basedir" contains: '
                             + ", ".join(map(os.path.basename, sequence_paths)) + "\n" + _DIRMSG)

        idx = np.arange(seqlen) * (dilation + 1)
        self.colorfiles, self.depthfiles, self.poses, self.framenames, self.timestamps = [], [], [], [], []
        for seq_path in sequence_paths:
            files = {}
            for name, needed in (("rgb.txt", True), ("depth.txt", True), ("groundtruth.txt", self.load_poses)):
                path = os.path.join(seq_path, name)
                if needed and not os.path.isfile(path):
                    what = 'poses file ("groundtruth.txt")' if name == "groundtruth.txt" else '"{}" file'.format(name)
                    raise ValueError("Missing {0} in {1}. ".format(what, path) + _DIRMSG)
                files[name] = path if needed else None
            seq_name = os.path.basename(seq_path)
            associations, seq_stamps = self._findAssociations(files["rgb.txt"], files["depth.txt"], files["groundtruth.txt"])
            for a in associations:
                if a[0][:3] != "rgb" or a[1][:5] != "depth":
                    raise ValueError("Incorrect reading from TUM associations")
            colors = [os.path.normpath(os.path.join(seq_path, a[0])) for a in associations]
            depths = [os.path.normpath(os.path.join(seq_path, a[1])) for a in associations]
            names = [seq_name.strip("/\\") + "/" + a[0][3:-4] for a in associations]
            for first in range(0, len(colors), stride):
                if first + idx[-1] >= len(colors):
                    break
                inds = first + idx
                self.colorfiles.append([colors[i] for i in inds])
                self.depthfiles.append([depths[i] for i in inds])
                self.framenames.append(", ".join(names[i] for i in inds))
                self.timestamps.append([seq_stamps[i] for i in inds])
                if self.load_poses:
                    self.poses.append([associations[i][2] for i in inds])
        self.num_sequences = len(self.colorfiles)

        intrinsics = torch.tensor([[525.0, 0, 319.5, 0], [0, 525.0, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]]).float()
        self.intrinsics = datautils.scale_intrinsics(intrinsics, self.height_downsample_ratio,
                                                     self.width_downsample_ratio).unsqueeze(0)
        self.scaling_factor = 5000.0

    def __len__(self):
        return self.num_sequences

    # ------------------------------------------------------------------ host path (reference semantics)
    def __getitem__(self, idx: int):
        color_seq, depth_seq = [], []
        for i in range(self.seqlen):
            color_seq.append(torch.from_numpy(self._preprocess_color(_imread(self.colorfiles[idx][i]).astype(float))))
            if self.return_depth:
                depth = _imread(self.depthfiles[idx][i]).astype(np.int64)
                depth_seq.append(torch.from_numpy(self._preprocess_depth(depth)))
        output = [torch.stack(color_seq, 0).float()]
        if self.return_depth:
            output.append(torch.stack(depth_seq, 0).float())
        if self.return_intrinsics:
            output.append(self.intrinsics)
        poses = self._homogenPoses(self.poses[idx]) if self.load_poses else None
        if self.return_pose:
            output.append(self._preprocess_poses(torch.stack([torch.from_numpy(p) for p in poses], 0).float()))
        if self.return_transform:
            output.append(torch.stack([torch.from_numpy(np.asarray(x)).float() for x in datautils.poses_to_transforms(poses)], 0))
        if self.return_names:
            output.append(self.framenames[idx])
        if self.return_timestamps:
            output.append("\n".join("rgb {} depth {} pose {}".format(*t) for t in self.timestamps[idx]))
        return tuple(output)

    def _preprocess_color(self, color: np.ndarray):
        color = _resize_bilinear(color, self.height, self.width)
        if self.normalize_color:
            color = datautils.normalize_image(color)
        if self.channels_first:
            color = datautils.channels_first(color)
        return color

    def _preprocess_depth(self, depth: np.ndarray):
        depth = np.expand_dims(_resize_nearest(depth.astype(float), self.height, self.width), -1)
        if self.channels_first:
            depth = datautils.channels_first(depth)
        return depth / self.scaling_factor

    def _preprocess_poses(self, poses: torch.Tensor):
        """Poses relative to the first frame of the sequence (first one = identity)."""
        return relative_transformation(poses[0].unsqueeze(0).repeat(poses.shape[0], 1, 1), poses)

    def _homogenPoses(self, poses_point_quaternion):
        return [datautils.pointquaternion_to_homogeneous(p) for p in poses_point_quaternion]

    def _findAssociations(self, rgb_text_file: str, depth_text_file: str, poses_text_file: Optional[str] = None,
                          max_difference: float = 0.02):
        """[(rgb path, depth path[, pose (7,)])] and [(rgb stamp, depth stamp, pose stamp)] of the frames whose
        time stamps pair up within `max_difference` seconds (reference :520-569)."""
        rgb = tumutils.read_file_list(rgb_text_file, self.start, self.end)
        depth = tumutils.read_file_list(depth_text_file)
        matches = tumutils.associate(rgb, depth, 0, float(max_difference))
        if poses_text_file is None:
            return ([(rgb[a][0], depth[b][0]) for a, b in matches], [(a, b, None) for a, b in matches])
        traj = tumutils.read_trajectory(poses_text_file, matrix=False)
        by_depth = {b: a for a, b in matches}
        triples = [(by_depth[d], d, p) for d, p in tumutils.associate(by_depth, traj, 0, float(max_difference))]
        return ([(rgb[a][0], depth[d][0], np.array(traj[p], dtype=np.float32)) for a, d, p in triples], list(triples))

    # ------------------------------------------------------------------ device path
    def load_rgbdimages(self, idx: int, device: Union[str, torch.device] = "cuda:0"):
        """Sequence `idx` as an RGBDImages (1, L, H, W, C) on `device`: raw frames uploaded as uint8 / uint16 and
        converted by the HIP kernel (scale, resize, normalise); poses relative to the first frame."""
        from .. import ops
        from ..structures.rgbdimages import RGBDImages

        rgb_raw = torch.from_numpy(np.stack([_imread(p) for p in self.colorfiles[idx]])).to(device)
        depth_raw = torch.from_numpy(np.stack([_imread(p) for p in self.depthfiles[idx]]).view(np.int16)).to(device)
        depth, rgb = ops.frames_from_raw(depth_raw, rgb_raw, self.height, self.width, self.scaling_factor, self.normalize_color)
        poses = None
        if self.load_poses:
            poses = self._preprocess_poses(torch.stack([torch.from_numpy(p) for p in self._homogenPoses(self.poses[idx])],
                                                       0).float()).unsqueeze(0).to(device)
        frames = RGBDImages(rgb.unsqueeze(0), depth.unsqueeze(0), self.intrinsics.unsqueeze(0).to(device), poses)
        return frames.to_channels_first() if self.channels_first else frames
