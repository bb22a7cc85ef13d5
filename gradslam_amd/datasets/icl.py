"""ICL-NUIM living-room trajectories (reference datasets/icl.py:18-572): same constructor and `__getitem__` tuple
(colour, depth, intrinsics, poses, transforms, names; each optional) as the reference.  Folder layout
`living_room_traj{N}_frei_png/{depth/, rgb/, associations.txt, livingRoom{N}n.gt.sim}`; `associations.txt` lines
are `id depth/ID.png id rgb/ID.png`; the pose file holds one 3x4 matrix per frame in blocks of four lines;
intrinsics fx=481.2, fy=-480, cx=319.5, cy=239.5 scaled to the requested size; depth = png / 5000."""
import os
import warnings
from typing import Optional, Union

import numpy as np
import torch

from . import datautils
from ._base import SequenceDataset

__all__ = ["ICL"]

_NAMING = "living_room_trajX_frei_png"


class ICL(SequenceDataset):
    orthogonal_rotations_default = False  # reference :515-533

    def __init__(self, basedir: str, trajectories: Union[tuple, str, None] = None, seqlen: int = 4,
                 dilation: Optional[int] = None, stride: Optional[int] = None, start: Optional[int] = None,
                 end: Optional[int] = None, height: int = 480, width: int = 640, channels_first: bool = False,
                 normalize_color: bool = False, *, return_depth: bool = True, return_intrinsics: bool = True,
                 return_pose: bool = True, return_transform: bool = True, return_names: bool = True):
        super().__init__()
        basedir = os.path.normpath(basedir)
        self._init_common(seqlen, dilation, stride, start, end, height, width, channels_first, normalize_color, return_depth,
                          return_intrinsics, return_pose, return_transform, return_names)

        def well_named(d):
            return d.startswith("living_room_traj") and d.endswith("_frei_png") and d[len("living_room_traj"):-len("_frei_png")].isdigit()

        available = sorted(f for f in os.listdir(basedir) if os.path.isdir(os.path.join(basedir, f)) and well_named(f))
        if len(available) == 0:
            raise ValueError("basedir ({0}) should contain trajectory folders with the following naming convention: "
                             '"{1}". Found no such folder.'.format(basedir, _NAMING))
        if isinstance(trajectories, str):
            if not os.path.isfile(trajectories):
                raise ValueError("incorrect filename: {} doesn't exist".format(trajectories))
            with open(trajectories, "r") as f:
                trajectories = tuple(t for t in f.read().split("\n") if t)
        elif not (trajectories is None or isinstance(trajectories, tuple)):
            raise TypeError('"trajectories" should either be path to .txt file or tuple of trajectory names or None, but was '
                            "of type {0} instead".format(type(trajectories)))
        if isinstance(trajectories, tuple):
            if len(trajectories) == 0:
                raise ValueError('"trajectories" must have atleast one element. Got len(trajectories)=0')
            for t in trajectories:
                if not well_named(t):
                    raise ValueError('"trajectories" should only contain trajectory folder names of the following convention: '
                                     '"{0}". It contained: {1}.'.format(_NAMING, t))
            missing = [t for t in trajectories if t not in available]
            if missing:
                raise ValueError('"trajectories" contains trajectories not available in basedir:\ntrajectories contains: '
                                 + ", ".join(trajectories) + "\nbasedir contains: " + ", ".join(available))
        chosen = [t for t in available if trajectories is None or t in trajectories]

        self.colorfiles, self.depthfiles, self.posemetas, self.framenames = [], [], [], []
        for name in chosen:
            tdir = os.path.join(basedir, name)
            assoc = os.path.join(tdir, "associations.txt")
            if not os.path.isfile(assoc):
                raise ValueError('Missing associations file ("associations.txt") in {0}. '.format(tdir))
            posesfile, n_pose_lines = None, 0
            if self.load_poses:
                num = name[len("living_room_traj"):].split("_")[0]
                posesfile = os.path.join(tdir, "livingRoom{0}n.gt.sim".format(num))
                if not os.path.isfile(posesfile):
                    raise ValueError('Missing ground truth poses file ("{0}") in {1}. '.format(posesfile, basedir))
                with open(posesfile, "r") as f:
                    n_pose_lines = sum(1 for _ in f)
            with open(assoc, "r") as f:
                lines = [ln for ln in f.readlines() if ln.strip()]
            stop = len(lines) if self.end is None else self.end
            if stop > len(lines):
                warnings.warn("end was larger than number of frames in trajectory: {0} > {1} (trajectory: {2})".format(
                    stop, len(lines), name))
            if name == "living_room_traj0_frei_png":
                lines = lines[:-1]  # traj0's pose file is one pose short (reference :313-315)
            lines = lines[self.start:stop]
            colors, depths, names, pose_lines = [], [], [], []
            for k, line in enumerate(lines):
                tok = line.strip().split()
                if tok[3][:3] != "rgb" or tok[1][:5] != "depth":
                    raise ValueError("incorrect reading from ICL associations")
                colors.append(os.path.normpath(os.path.join(tdir, tok[3])))
                depths.append(os.path.normpath(os.path.join(tdir, tok[1])))
                names.append(os.path.join(name, tok[1][6:].split(".")[0]))
                if self.load_poses:
                    if k * 4 > n_pose_lines:
                        raise ValueError('{0}th pose should start from line {1} of file "{2}", but said file has only {3} '
                                         "lines.".format(k, k * 4, os.path.join(*posesfile.split(os.sep)[-2:]), n_pose_lines))
                    pose_lines.append(k * 4)
            for inds in self._windows(len(colors)):
                self.colorfiles.append([colors[i] for i in inds])
                self.depthfiles.append([depths[i] for i in inds])
                self.framenames.append(", ".join(names[i] for i in inds))
                if self.load_poses:
                    self.posemetas.append({"file": posesfile, "line_nums": [pose_lines[i] for i in inds]})
        self.num_sequences = len(self.colorfiles)

        intrinsics = torch.tensor([[481.20, 0, 319.5, 0], [0, -480.0, 239.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]]).float()
        self.intrinsics = datautils.scale_intrinsics(intrinsics, self.height_downsample_ratio,
                                                     self.width_downsample_ratio).unsqueeze(0)
        self.scaling_factor = 5000.0

    def _sequence_poses(self, idx: int):
        meta = self.posemetas[idx]
        return self._loadPoses(meta["file"], meta["line_nums"])

    def _loadPoses(self, pose_path, start_lines):
        """The 3x4 matrices that start at the given line numbers, as float32 4x4 (reference :535-572)."""
        with open(pose_path, "r") as f:
            lines = f.readlines()
        wanted = set(start_lines)
        poses = []
        for first in sorted(wanted):
            rows = []
            for line in lines[first:first + 3]:
                tok = line.strip().split()
                if len(tok) != 4:
                    raise ValueError("Faulty poses file: Expected line {0} of the poses file {1} to contain pose matrix values, "
                                     "but it didn't. You can download 'Global_RT_Trajectory_GT' from here:\n"
                                     "https://www.doc.ic.ac.uk/~ahanda/VaFRIC/iclnuim.html".format(first, pose_path))
                rows.append(tok)
            if len(rows) != 3:
                raise ValueError("Faulty poses file: pose at line {0} of {1} is incomplete".format(first, pose_path))
            rows.append([0.0, 0.0, 0.0, 1.0])
            poses.append(np.array(rows, dtype=np.float32))
        return poses
