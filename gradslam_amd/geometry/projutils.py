"""Projection helpers the containers expose (reference geometry/projutils.py:10-43, :46-89, :92-238,
:241-402, :405-450).  Small tensor algebra on whatever device the inputs live on; the hot path itself uses
the fused HIP kernels (project.hip, maps.hip), not these."""
from typing import Optional

import torch

__all__ = ["homogenize_points", "unhomogenize_points", "project_points", "unproject_points", "inverse_intrinsics"]


def homogenize_points(pts: torch.Tensor) -> torch.Tensor:
    if not isinstance(pts, torch.Tensor):
        raise TypeError("Expected input type torch.Tensor. Got {} instead".format(type(pts)))
    if pts.dim() < 2:
        raise ValueError("Input tensor must have at least 2 dimensions. Got {} instad.".format(pts.dim()))
    return torch.nn.functional.pad(pts, (0, 1), "constant", 1.0)


def unhomogenize_points(pts: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """(N,*,K) -> (N,*,K-1): divide by the last coordinate; |w| <= eps (a point at infinity) divides by 1."""
    if not isinstance(pts, torch.Tensor):
        raise TypeError("Expected input type torch.Tensor. Instead got {}".format(type(pts)))
    if pts.dim() < 2:
        raise ValueError("Input tensor must have at least 2 dimensions. Got {} instad.".format(pts.dim()))
    w = pts[..., -1:]
    return pts[..., :-1] * torch.where(w.abs() > eps, w.reciprocal(), torch.ones_like(w))


def _broadcast_mat(mat: torch.Tensor, homo: torch.Tensor) -> torch.Tensor:
    # an unbatched matrix is shared by every leading dimension of the points; the result gets a points axis
    if mat.dim() == 2 and homo.dim() > 2:
        mat = mat.reshape((1,) * (homo.dim() - 2) + tuple(mat.shape))
    return mat.unsqueeze(-3) if homo.dim() > 2 else mat.unsqueeze(0)


def project_points(cam_coords: torch.Tensor, proj_mat: torch.Tensor, eps: Optional[float] = 1e-6) -> torch.Tensor:
    """(N,*,3|4) points x (*,4,4) projection -> (N,*,2) pixel coordinates; z == 0 divides by 1."""
    if not torch.is_tensor(cam_coords):
        raise TypeError("Expected input cam_coords to be of type torch.Tensor. Got {0} instead.".format(type(cam_coords)))
    if not torch.is_tensor(proj_mat):
        raise TypeError("Expected input proj_mat to be of type torch.Tensor. Got {0} instead.".format(type(proj_mat)))
    if cam_coords.dim() < 2:
        raise ValueError("Input cam_coords must have at least 2 dims. Got {0} instead.".format(cam_coords.dim()))
    if cam_coords.shape[-1] not in (3, 4):
        raise ValueError("Input cam_coords must have shape (*, 3), or (*, 4). Got {0} instead.".format(cam_coords.shape))
    if proj_mat.dim() < 2:
        raise ValueError("Input proj_mat must have at least 2 dims. Got {0} instead.".format(proj_mat.dim()))
    if proj_mat.shape[-1] != 4 or proj_mat.shape[-2] != 4:
        raise ValueError("Input proj_mat must have shape (*, 4, 4). Got {0} instead.".format(proj_mat.shape))
    if proj_mat.dim() > 2 and proj_mat.dim() != cam_coords.dim():
        raise ValueError("Input proj_mat must either have 2 dimensions, or have equal number of dimensions to "
                         "cam_coords. Got {0} instead.".format(proj_mat.dim()))
    if proj_mat.dim() > 2 and proj_mat.shape[0] != cam_coords.shape[0]:
        raise ValueError("Batch sizes of proj_mat and cam_coords do not match. Shapes: {0} and {1} respectively.".format(
            proj_mat.shape, cam_coords.shape))
    homo = homogenize_points(cam_coords) if cam_coords.shape[-1] == 3 else cam_coords
    q = torch.matmul(_broadcast_mat(proj_mat, homo), homo.unsqueeze(-1)).squeeze(-1)
    z = q[..., 2]
    safe = torch.where(z != 0, z, torch.ones_like(z))
    return torch.stack((q[..., 0] / safe, q[..., 1] / safe), dim=-1)


def unproject_points(pixel_coords: torch.Tensor, intrinsics_inv: torch.Tensor, depths: torch.Tensor) -> torch.Tensor:
    """(N,*,2|3) pixels x (*,3,3) inverse intrinsics x (N,*) depths -> (N,*,3) camera-frame points."""
    if not torch.is_tensor(pixel_coords):
        raise TypeError("Expected input pixel_coords to be of type torch.Tensor. Got {0} instead.".format(type(pixel_coords)))
    if not torch.is_tensor(intrinsics_inv):
        raise TypeError("Expected intrinsics_inv to be of type torch.Tensor. Got {0} instead.".format(type(intrinsics_inv)))
    if not torch.is_tensor(depths):
        raise TypeError("Expected depth to be of type torch.Tensor. Got {0} instead.".format(type(depths)))
    if pixel_coords.dim() < 2:
        raise ValueError("Input pixel_coords must have at least 2 dims. Got {0} instead.".format(pixel_coords.dim()))
    if pixel_coords.shape[-1] not in (2, 3):
        raise ValueError("Input pixel_coords must have shape (*, 2), or (*, 3). Got {0} instead.".format(pixel_coords.shape))
    if intrinsics_inv.dim() < 2:
        raise ValueError("Input intrinsics_inv must have at least 2 dims. Got {0} instead.".format(intrinsics_inv.dim()))
    if intrinsics_inv.shape[-1] != 3 or intrinsics_inv.shape[-2] != 3:
        raise ValueError("Input intrinsics_inv must have shape (*, 3, 3). Got {0} instead.".format(intrinsics_inv.shape))
    if intrinsics_inv.dim() > 2 and intrinsics_inv.dim() != pixel_coords.dim():
        raise ValueError("Input intrinsics_inv must either have 2 dimensions, or have equal number of dimensions to "
                         "pixel_coords. Got {0} instead.".format(intrinsics_inv.dim()))
    if intrinsics_inv.dim() > 2 and intrinsics_inv.shape[0] != pixel_coords.shape[0]:
        raise ValueError("Batch sizes of intrinsics_inv and pixel_coords do not match. Shapes: {0} and {1} "
                         "respectively.".format(intrinsics_inv.shape, pixel_coords.shape))
    if pixel_coords.shape[:-1] != depths.shape:
        raise ValueError("Input pixel_coords and depths must have the same shape for all dimensions except the last. "
                         " Got {0} and {1} respectively.".format(pixel_coords.shape, depths.shape))
    homo = homogenize_points(pixel_coords) if pixel_coords.shape[-1] == 2 else pixel_coords
    rays = torch.matmul(_broadcast_mat(intrinsics_inv, homo), homo.unsqueeze(-1)).squeeze(-1)
    return rays * depths.unsqueeze(-1)


def inverse_intrinsics(K: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """Closed-form inverse of a pinhole matrix; note the +eps on the focal lengths."""
    if not torch.is_tensor(K):
        raise TypeError("Expected K to be of type torch.Tensor. Got {0} instead.".format(type(K)))
    if K.dim() < 2:
        raise ValueError("Input K must have at least 2 dims. Got {0} instead.".format(K.dim()))
    if not ((K.shape[-1] == 3 and K.shape[-2] == 3) or (K.shape[-1] == 4 and K.shape[-2] == 4)):
        raise ValueError("Input K must have shape (*, 4, 4) or (*, 3, 3). Got {0} instead.".format(K.shape))
    out = torch.zeros_like(K)
    fx, fy = K[..., 0, 0] + eps, K[..., 1, 1] + eps
    out[..., 0, 0] = 1.0 / fx
    out[..., 1, 1] = 1.0 / fy
    out[..., 0, 2] = -1.0 * K[..., 0, 2] / fx
    out[..., 1, 2] = -1.0 * K[..., 1, 2] / fy
    out[..., 2, 2] = 1
    out[..., -1, -1] = 1
    return out
