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

from . import datautils, tumutils
from ._base import SequenceDataset

__all__ = ["TUM"]

_DIRMSG = ("TUM folder should look something like:\n\n| ├── basedir\n| │   ├── rgbd_dataset_freiburgX_NAME\n"
           "| │   │   ├── depth/\n| │   │   ├── rgb/\n| │   │   ├── accelerometer.txt\n| │   │   └── depth.txt\n"
           "| │   │   └── groundtruth.txt\n| │   │   └── rgb.txt\n| │   ├── ...")


class TUM(SequenceDataset):
    def __init__(self, basedir: str, sequences: Union[tuple, str, None] = None, seqlen: int = 4,
                 dilation: Optional[int] = None, stride: Optional[int] = None, start: Optional[int] = None,
                 end: Optional[int] = None, height: int = 480, width: int = 640, channels_first: bool = False,
                 normalize_color: bool = False, *, return_depth: bool = True, return_intrinsics: bool = True,
                 return_pose: bool = True, return_transform: bool = True, return_names: bool = True,
                 return_timestamps: bool = True):
        super().__init__()
        basedir = os.path.normpath(basedir)
        self._init_common(seqlen, dilation, stride, start, end, height, width, channels_first, normalize_color, return_depth,
                          return_intrinsics, return_pose, return_transform, return_names)
        self.return_timestamps = return_timestamps

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
            for inds in self._windows(len(colors)):
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

    def _sequence_poses(self, idx: int):
        return self._homogenPoses(self.poses[idx])

    def _extra_outputs(self, idx: int) -> tuple:
        if not self.return_timestamps:
            return ()
        return ("\n".join("rgb {} depth {} pose {}".format(*t) for t in self.timestamps[idx]),)

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
