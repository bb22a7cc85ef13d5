"""ScanNet sequences (reference datasets/scannet.py:17-527): same constructor and `__getitem__` tuple (colour,
depth, intrinsics, poses, transforms, names, labels; each optional).  A sequence is described by a metadata file
`<seqmetadir>/sceneXXXX_XX-seq_Y.txt` whose lines read

    color <path> depth <path> pose <path> label-filt <path> ... intrinsic_depth <path>

(fields 0..7 and 14..15 are used, paths relative to `basedir`); poses and the depth intrinsics are 4x4 text
matrices; depth = png / 1000; labels are NYU40 ids, optionally folded onto the 20 ScanNet benchmark classes."""
import glob
import os
import re
from collections import OrderedDict
from typing import Optional, Union

import numpy as np
import torch

from . import datautils
from ._base import SequenceDataset, imread, resize_nearest

__all__ = ["Scannet", "get_color_encoding", "nyu40_to_scannet20"]

# NYU40 class names in id order (0 = unlabeled) and the ScanNet benchmark's colour of each (public label
# definitions: kaldir.vc.in.tum.de/scannet_benchmark/labelids_all.txt)
_NYU40 = ("unlabeled wall floor cabinet bed chair sofa table door window bookshelf picture counter blinds desk shelves "
          "curtain dresser pillow mirror floormat clothes ceiling books refrigerator television paper towel showercurtain "
          "box whiteboard person nightstand toilet sink lamp bathtub bag otherstructure otherfurniture otherprop").split()
_NYU40_RGB = ((0, 0, 0), (174, 199, 232), (152, 223, 138), (31, 119, 180), (255, 187, 120), (188, 189, 34), (140, 86, 75),
              (255, 152, 150), (214, 39, 40), (197, 176, 213), (148, 103, 189), (196, 156, 148), (23, 190, 207),
              (178, 76, 76), (247, 182, 210), (66, 188, 102), (219, 219, 141), (140, 57, 197), (202, 185, 52),
              (51, 176, 203), (200, 54, 131), (92, 193, 61), (78, 71, 183), (172, 114, 82), (255, 127, 14), (91, 163, 138),
              (153, 98, 156), (140, 153, 101), (158, 218, 229), (100, 125, 154), (178, 127, 135), (120, 185, 128),
              (146, 111, 194), (44, 160, 44), (112, 128, 144), (96, 207, 209), (227, 119, 194), (213, 92, 176),
              (94, 106, 211), (82, 84, 163), (100, 85, 144))
# the NYU40 ids that make up the 20-class benchmark, in benchmark order (labelids.txt)
_SCANNET20_IDS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39)


def get_color_encoding(seg_classes: str) -> OrderedDict:
    """{class name: (r, g, b)} of the `"nyu40"` or `"scannet20"` palette (reference :359-438)."""
    if seg_classes.lower() == "nyu40":
        return OrderedDict(zip(_NYU40, _NYU40_RGB))
    if seg_classes.lower() == "scannet20":
        ids = (0,) + _SCANNET20_IDS
        return OrderedDict((_NYU40[i], _NYU40_RGB[i]) for i in ids)
    return None


def nyu40_to_scannet20(label: np.ndarray) -> np.ndarray:
    """NYU40 ids -> contiguous 0..20 (classes outside the benchmark become 0), in place like the reference
    (:441-488); done with one lookup instead of 29 masked assignments."""
    lut = np.zeros(256, dtype=label.dtype)
    for new, old in enumerate(_SCANNET20_IDS, start=1):
        lut[old] = new
    ids = np.arange(256)
    lut[41:] = ids[41:].astype(label.dtype)  # ids beyond NYU40 pass through untouched, as in the reference
    label[...] = lut[label]
    return label


class Scannet(SequenceDataset):
    def __init__(self, basedir: str, seqmetadir: str, scenes: Union[tuple, str, None], start: Optional[int] = 0,
                 end: Optional[int] = -1, height: int = 480, width: int = 640, seg_classes: str = "scannet20",
                 channels_first: bool = False, normalize_color: bool = False, *, return_depth: bool = True,
                 return_intrinsics: bool = True, return_pose: bool = True, return_transform: bool = True,
                 return_names: bool = True, return_labels: bool = True):
        torch.utils.data.Dataset.__init__(self)
        basedir = os.path.normpath(basedir)
        self.height, self.width = height, width
        self.height_downsample_ratio = float(height) / 480
        self.width_downsample_ratio = float(width) / 640
        self.seg_classes = seg_classes
        self.channels_first, self.normalize_color = channels_first, normalize_color
        self.return_depth, self.return_intrinsics = return_depth, return_intrinsics
        self.return_pose, self.return_transform = return_pose, return_transform
        self.return_names, self.return_labels = return_names, return_labels
        self.load_poses = return_pose or return_transform
        self.color_encoding = get_color_encoding(self.seg_classes)
        self.start, self.end = start, end
        full_sequence = self.end == -1
        if start < 0:
            raise ValueError("Start frame cannot be less than 0.")
        if not (end == -1 or end > start):
            raise ValueError("End frame ({}) should be equal to -1 or greater than start ({})".format(end, start))
        self.seqlen = self.end - self.start
        if isinstance(scenes, str):
            if not os.path.isfile(scenes):
                raise ValueError("incorrect filename: {} doesn't exist".format(scenes))
            with open(scenes, "r") as f:
                scenes = tuple(f.read().split("\n"))
        elif not (scenes is None or isinstance(scenes, tuple)):
            raise TypeError("scenes should either be path to split.txt or tuple of scenes or None, but was of type %r instead"
                            % type(scenes))

        natural = lambda s: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]  # natsort's ordering
        self.colorfiles, self.depthfiles, self.posefiles, self.labelfiles = [], [], [], []
        self.intrinsicsfiles, self.seqnames = [], []
        for meta in sorted(glob.glob(os.path.join(seqmetadir, "*.txt")), key=natural):
            if scenes is not None and os.path.basename(meta).split("-")[0] not in scenes:
                continue
            with open(meta, "r") as f:
                lines = f.readlines()
            if full_sequence:
                self.end = len(lines)
                self.seqlen = self.end - self.start
            if self.seqlen > len(lines):
                raise ValueError("sequence length can't be larger than dataset sequence length but it was: %r > %r"
                                 % (self.seqlen, len(lines)))
            cols = {"color": [], "depth": [], "pose": [], "label-filt": [], "intrinsic_depth": []}
            for line in lines[self.start:self.end]:
                tok = line.strip().split()
                for key, at in (("color", 0), ("depth", 2), ("pose", 4), ("label-filt", 6), ("intrinsic_depth", 14)):
                    if len(tok) <= at + 1 or tok[at] != key:
                        raise ValueError("incorrect reading from scannet metadata")
                    cols[key].append(os.path.join(basedir, tok[at + 1]))
            self.colorfiles.append(cols["color"])
            self.depthfiles.append(cols["depth"])
            self.posefiles.append(cols["pose"])
            self.labelfiles.append(cols["label-filt"])
            self.intrinsicsfiles.append(cols["intrinsic_depth"][0])
            self.seqnames.append(os.path.basename(meta).split(".")[0])
        self.framenames = self.seqnames
        self.num_sequences = len(self.colorfiles)
        self.scaling_factor = 1000.0

    # per-sequence intrinsics (the other loaders have one camera)
    def _sequence_intrinsics(self, idx: int) -> torch.Tensor:
        return torch.from_numpy(self._preprocess_intrinsics(np.loadtxt(self.intrinsicsfiles[idx]).astype(float))).float()

    def _preprocess_intrinsics(self, intrinsics):
        scaled = datautils.scale_intrinsics(intrinsics, self.height_downsample_ratio, self.width_downsample_ratio)
        return scaled.unsqueeze(0) if torch.is_tensor(scaled) else np.expand_dims(scaled, 0)

    def _sequence_poses(self, idx: int):
        return [np.loadtxt(p).astype(float) for p in self.posefiles[idx]]

    def _preprocess_label(self, label: np.ndarray):
        label = resize_nearest(label, self.height, self.width).copy()
        if self.seg_classes.lower() == "scannet20":
            label = nyu40_to_scannet20(label)
        return np.expand_dims(label, -1)

    def __getitem__(self, idx: int):
        color_seq, depth_seq, label_seq = [], [], []
        for i in range(self.seqlen):
            color_seq.append(torch.from_numpy(self._preprocess_color(imread(self.colorfiles[idx][i]).astype(float))))
            if self.return_depth:
                depth_seq.append(torch.from_numpy(self._preprocess_depth(imread(self.depthfiles[idx][i]).astype(np.int64))))
            if self.return_labels:
                from PIL import Image

                with Image.open(self.labelfiles[idx][i]) as im:
                    label_seq.append(torch.from_numpy(self._preprocess_label(np.asarray(im).astype(np.uint8))))
        output = [torch.stack(color_seq, 0).float()]
        if self.return_depth:
            output.append(torch.stack(depth_seq, 0).float())
        if self.return_intrinsics:
            output.append(self._sequence_intrinsics(idx))
        poses = self._sequence_poses(idx) if self.load_poses else None
        if self.return_pose:
            output.append(self._preprocess_poses(torch.stack([torch.from_numpy(p) for p in poses], 0).float()))
        if self.return_transform:
            from .datautils import poses_to_transforms

            output.append(torch.stack([torch.from_numpy(np.asarray(x)).float() for x in poses_to_transforms(poses)], 0))
        if self.return_names:
            output.append(self.seqnames[idx])
        if self.return_labels:
            output.append(torch.stack(label_seq, 0).float())
        return tuple(output)

    def load_rgbdimages(self, idx: int, device="cuda:0"):
        """Sequence `idx` on the device (raw uint8 / uint16 frames converted by the HIP kernel), with its own
        intrinsics; labels are not part of an RGBDImages."""
        from .. import ops
        from ..structures.rgbdimages import RGBDImages

        rgb_raw = torch.from_numpy(np.stack([imread(p) for p in self.colorfiles[idx]])).to(device)
        depth_raw = torch.from_numpy(np.stack([imread(p) for p in self.depthfiles[idx]]).view(np.int16)).to(device)
        depth, rgb = ops.frames_from_raw(depth_raw, rgb_raw, self.height, self.width, self.scaling_factor, self.normalize_color)
        poses = None
        if self.load_poses:
            poses = self._preprocess_poses(torch.stack([torch.from_numpy(p) for p in self._sequence_poses(idx)], 0).float())
            poses = poses.unsqueeze(0).to(device)
        frames = RGBDImages(rgb.unsqueeze(0), depth.unsqueeze(0), self._sequence_intrinsics(idx).unsqueeze(0).to(device), poses)
        return frames.to_channels_first() if self.channels_first else frames
