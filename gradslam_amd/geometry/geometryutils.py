"""Rigid-body helpers used by the SLAM drivers (reference geometry/geometryutils.py:205-301,
:413-478, :576-608, :737-794 and the two kornia.geometry.linalg functions the reference imports).
O(1) 4x4 algebra in torch; point clouds go through the HIP transform kernel."""
from typing import Optional

import torch

__all__ = ["create_meshgrid", "transform_pointcloud", "inverse_transformation", "compose_transformations",
           "relative_transformation"]


def create_meshgrid(height: int, width: int, normalized_coords: Optional[bool] = True) -> torch.Tensor:
    """(1, H, W, 2) grid of (row, col) coordinates (float32 linspace, like the reference)."""
    if normalized_coords:
        rows, cols = torch.linspace(-1, 1, height), torch.linspace(-1, 1, width)
    else:
        rows, cols = torch.linspace(0, height - 1, height), torch.linspace(0, width - 1, width)
    rr, cc = torch.meshgrid([rows, cols], indexing="ij")
    return torch.stack((rr, cc), dim=-1).unsqueeze(0)


def transform_pointcloud(pointcloud: torch.Tensor, transform: torch.Tensor) -> torch.Tensor:
    """(N,3) points, (4,4) rigid transform -> R p + t.  HIP tensors run the transform kernel."""
    if not torch.is_tensor(pointcloud):
        raise TypeError("pointcloud should be tensor, but was %r instead" % type(pointcloud))
    if not torch.is_tensor(transform):
        raise TypeError("transform should be tensor, but was %r instead" % type(transform))
    if not pointcloud.ndim == 2:
        raise ValueError("pointcloud should have ndim of 2, but had {} instead.".format(pointcloud.ndim))
    if not pointcloud.shape[1] == 3:
        raise ValueError("pointcloud.shape[1] should be 3 (x, y, z), but was {} instead.".format(pointcloud.shape[1]))
    if not transform.shape[-2:] == (4, 4):
        raise ValueError("transform should be of shape (4, 4), but was {} instead.".format(transform.shape))
    from .. import ops

    return ops.transform_points(pointcloud, transform)


def inverse_transformation(trans_12: torch.Tensor) -> torch.Tensor:
    """(R, t) -> (R^T, -R^T t) for (*,4,4) rigid transforms."""
    if not torch.is_tensor(trans_12):
        raise TypeError("Input type is not a torch.Tensor. Got {}".format(type(trans_12)))
    if trans_12.dim() not in (2, 3) or trans_12.shape[-2:] != (4, 4):
        raise ValueError("Input size must be a Nx4x4 or 4x4. Got {}".format(trans_12.shape))
    Rt = trans_12[..., :3, :3].transpose(-1, -2)
    out = torch.zeros_like(trans_12)
    out[..., :3, :3] = out[..., :3, :3] + Rt
    out[..., :3, 3:4] = out[..., :3, 3:4] + torch.matmul(-Rt, trans_12[..., :3, 3:4])
    out[..., 3, 3] = out[..., 3, 3] + 1.0
    return out


def compose_transformations(trans_01: torch.Tensor, trans_12: torch.Tensor) -> torch.Tensor:
    """T_02 = T_01 . T_12 for (*,4,4) rigid transforms (bottom row forced to [0,0,0,1])."""
    if not torch.is_tensor(trans_01) or not torch.is_tensor(trans_12):
        raise TypeError("Inputs must be torch.Tensors. Got {} and {}".format(type(trans_01), type(trans_12)))
    if trans_01.dim() not in (2, 3) or trans_01.shape[-2:] != (4, 4):
        raise ValueError("Input trans_01 must be a of the shape Nx4x4 or 4x4. Got {}".format(trans_01.shape))
    if trans_12.dim() not in (2, 3) or trans_12.shape[-2:] != (4, 4):
        raise ValueError("Input trans_12 must be a of the shape Nx4x4 or 4x4. Got {}".format(trans_12.shape))
    if trans_01.dim() != trans_12.dim():
        raise ValueError("Input number of dims must match. Got {} and {}".format(trans_01.dim(), trans_12.dim()))
    R = torch.matmul(trans_01[..., :3, :3], trans_12[..., :3, :3])
    t = torch.matmul(trans_01[..., :3, :3], trans_12[..., :3, 3:4]) + trans_01[..., :3, 3:4]
    out = torch.zeros_like(trans_01)
    out[..., :3, :3] = out[..., :3, :3] + R
    out[..., :3, 3:4] = out[..., :3, 3:4] + t
    out[..., 3, 3] = out[..., 3, 3] + 1.0
    return out


def relative_transformation(trans_01: torch.Tensor, trans_02: torch.Tensor, orthogonal_rotations: bool = False):
    """T_12 = inv(T_01) . T_02 (reference geometry/geometryutils.py:413-478)."""
    if not torch.is_tensor(trans_01) or not torch.is_tensor(trans_02):
        raise TypeError("Inputs must be torch.Tensors. Got {} and {}".format(type(trans_01), type(trans_02)))
    if trans_01.dim() not in (2, 3) or trans_01.shape[-2:] != (4, 4):
        raise ValueError("Input must be a of the shape Nx4x4 or 4x4. Got {}".format(trans_01.shape))
    if trans_02.dim() not in (2, 3) or trans_02.shape[-2:] != (4, 4):
        raise ValueError("Input must be a of the shape Nx4x4 or 4x4. Got {}".format(trans_02.shape))
    if trans_01.dim() != trans_02.dim():
        raise ValueError("Input number of dims must match. Got {} and {}".format(trans_01.dim(), trans_02.dim()))
    inv = inverse_transformation(trans_01) if orthogonal_rotations else torch.inverse(trans_01)
    return compose_transformations(inv, trans_02)
