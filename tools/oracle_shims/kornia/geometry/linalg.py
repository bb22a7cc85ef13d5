"""Stand-in for the two kornia functions the reference path calls (4x4 rigid algebra)."""
import torch


def inverse_transformation(trans_12: torch.Tensor) -> torch.Tensor:
    rmat_12 = trans_12[..., :3, :3]
    tvec_12 = trans_12[..., :3, 3:4]
    rmat_21 = rmat_12.transpose(-1, -2)
    tvec_21 = torch.matmul(-rmat_21, tvec_12)
    trans_21 = torch.zeros_like(trans_12)
    trans_21[..., :3, :3] = trans_21[..., :3, :3] + rmat_21
    trans_21[..., :3, 3:4] = trans_21[..., :3, 3:4] + tvec_21
    trans_21[..., 3, 3] = trans_21[..., 3, 3] + 1.0
    return trans_21


def compose_transformations(trans_01: torch.Tensor, trans_12: torch.Tensor) -> torch.Tensor:
    rmat_01 = trans_01[..., :3, :3]
    rmat_12 = trans_12[..., :3, :3]
    tvec_01 = trans_01[..., :3, 3:4]
    tvec_12 = trans_12[..., :3, 3:4]
    rmat_02 = torch.matmul(rmat_01, rmat_12)
    tvec_02 = torch.matmul(rmat_01, tvec_12) + tvec_01
    trans_02 = torch.zeros_like(trans_01)
    trans_02[..., :3, :3] = trans_02[..., :3, :3] + rmat_02
    trans_02[..., :3, 3:4] = trans_02[..., :3, 3:4] + tvec_02
    trans_02[..., 3, 3] = trans_02[..., 3, 3] + 1.0
    return trans_02
