"""Array helpers of the dataset front-end (reference datasets/datautils.py:19-263): pure host algebra on
numpy arrays or torch tensors, same names / argument meaning / error behaviour."""
import copy
import warnings
from collections import OrderedDict
from typing import List, Union

import numpy as np
import torch

__all__ = ["normalize_image", "channels_first", "scale_intrinsics", "pointquaternion_to_homogeneous",
           "poses_to_transforms", "create_label_image"]


def normalize_image(rgb: Union[torch.Tensor, np.ndarray]):
    """[0, 255] -> [0, 1] (reference :19-37)."""
    if torch.is_tensor(rgb):
        return rgb.float() / 255
    if isinstance(rgb, np.ndarray):
        return rgb.astype(float) / 255
    raise TypeError("Unsupported input rgb type: %r" % type(rgb))


def channels_first(rgb: Union[torch.Tensor, np.ndarray]):
    """(*, H, W, C) -> (*, C, H, W), contiguous (reference :40-70)."""
    if not (isinstance(rgb, np.ndarray) or torch.is_tensor(rgb)):
        raise TypeError("Unsupported input rgb type {}".format(type(rgb)))
    if rgb.ndim < 3:
        raise ValueError("Input rgb must contain atleast 3 dims, but had {} dims.".format(rgb.ndim))
    if rgb.shape[-3] < rgb.shape[-1]:
        warnings.warn("Are you sure that the input is correct? Number of channels exceeds height of image: %r > %r"
                      % (rgb.shape[-1], rgb.shape[-3]))
    lead = list(range(rgb.ndim - 3))
    order = lead + [rgb.ndim - 1, rgb.ndim - 3, rgb.ndim - 2]
    if isinstance(rgb, np.ndarray):
        return np.ascontiguousarray(rgb.transpose(*order))
    return rgb.permute(*order).contiguous()


def scale_intrinsics(intrinsics: Union[np.ndarray, torch.Tensor], h_ratio: Union[float, int], w_ratio: Union[float, int]):
    """Intrinsics of a frame resized by (h_ratio, w_ratio): fx, cx scale with the width, fy, cy with the
    height (reference :73-117).  (*, 3, 3) or (*, 4, 4) in, float32 copy out."""
    if isinstance(intrinsics, np.ndarray):
        out = intrinsics.astype(np.float32).copy()
    elif torch.is_tensor(intrinsics):
        out = intrinsics.to(torch.float).clone()
    else:
        raise TypeError("Unsupported input intrinsics type {}".format(type(intrinsics)))
    if tuple(intrinsics.shape[-2:]) not in ((3, 3), (4, 4)):
        raise ValueError("intrinsics must have shape (*, 3, 3) or (*, 4, 4), but had shape {} instead".format(intrinsics.shape))
    if (intrinsics[..., -1, -1] != 1).any() or (intrinsics[..., 2, 2] != 1).any():
        warnings.warn("Incorrect intrinsics: intrinsics[..., -1, -1] and intrinsics[..., 2, 2] should be 1.")
    out[..., 0, 0] *= w_ratio
    out[..., 0, 2] *= w_ratio
    out[..., 1, 1] *= h_ratio
    out[..., 1, 2] *= h_ratio
    return out


def pointquaternion_to_homogeneous(pointquaternions: Union[np.ndarray, torch.Tensor], eps: float = 1e-12):
    """(tx, ty, tz, qx, qy, qz, qw) -> 4x4 [R | t] (reference :120-215): q is scaled to norm sqrt(2) so that
    the outer product q q^T holds the doubled products the rotation matrix is made of."""
    if not (isinstance(pointquaternions, np.ndarray) or torch.is_tensor(pointquaternions)):
        raise TypeError('"pointquaternions" must be of type "np.ndarray" or "torch.Tensor". Got {0}'.format(type(pointquaternions)))
    if not isinstance(eps, float):
        raise TypeError('"eps" must be of type "float". Got {0}.'.format(type(eps)))
    if pointquaternions.shape[-1] != 7:
        raise ValueError('"pointquaternions" must be of shape (*, 7). Got {0}.'.format(pointquaternions.shape))
    is_np = isinstance(pointquaternions, np.ndarray)
    pq = torch.from_numpy(np.ascontiguousarray(pointquaternions)) if is_np else pointquaternions
    t, q = pq[..., :3].float(), pq[..., 3:7].float()
    half_norm = (0.5 * (q ** 2).sum(-1, keepdim=True)) ** 0.5
    q = q / torch.clamp(half_norm, min=eps)
    o = q.unsqueeze(-1) * q.unsqueeze(-2)  # o[i][j] = q_i q_j, indices x, y, z, w
    T = torch.zeros((*pq.shape[:-1], 4, 4), dtype=torch.float32, device=pq.device)
    T[..., 0, 0] = 1.0 - (o[..., 1, 1] + o[..., 2, 2])
    T[..., 0, 1] = o[..., 0, 1] - o[..., 2, 3]
    T[..., 0, 2] = o[..., 0, 2] + o[..., 1, 3]
    T[..., 1, 0] = o[..., 0, 1] + o[..., 2, 3]
    T[..., 1, 1] = 1.0 - (o[..., 0, 0] + o[..., 2, 2])
    T[..., 1, 2] = o[..., 1, 2] - o[..., 0, 3]
    T[..., 2, 0] = o[..., 0, 2] - o[..., 1, 3]
    T[..., 2, 1] = o[..., 1, 2] + o[..., 0, 3]
    T[..., 2, 2] = 1.0 - (o[..., 0, 0] + o[..., 1, 1])
    T[..., :3, 3] = t
    T[..., 3, 3] = 1.0
    return T.numpy() if is_np else T


def poses_to_transforms(poses: Union[np.ndarray, List[np.ndarray]]):
    """Frame-to-frame transforms inv(P[i-1]) P[i], identity for the first frame (reference :218-239)."""
    out = copy.deepcopy(poses)
    for i in range(len(poses)):
        out[i] = np.eye(4) if i == 0 else np.linalg.inv(poses[i - 1]).dot(poses[i])
    return out


def create_label_image(prediction: np.ndarray, color_palette: OrderedDict):
    """(H, W) class indices -> (H, W, 3) uint8 colours (reference :242-263)."""
    label = np.zeros((prediction.shape[0], prediction.shape[1], 3), dtype=np.uint8)
    for idx, color in enumerate(color_palette):
        label[prediction == idx] = color
    return label
