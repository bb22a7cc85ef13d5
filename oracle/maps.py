"""Depth -> vertex / normal maps (local + global), channels-last.  fp32 torch CPU ops."""
import torch

from .geometry import inverse_intrinsics, pixel_grid


def vertex_map(depth: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """depth (B,L,H,W,1), K (B,1,4,4) -> (B,L,H,W,3).  reference
    structures/rgbdimages.py:643-679 (einsum then *depth then *valid)."""
    B, L, H, W, _ = depth.shape
    pix = pixel_grid(B, L, H, W).to(depth.device)
    Kinv = inverse_intrinsics(K)[..., :3, :3].repeat(1, L, 1, 1)
    V = torch.einsum("bsjc,bshwc->bshwj", Kinv, pix) * depth
    return V * (depth > 0).to(V.dtype)


def normal_map(V: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
    """Forward-difference cross-product normals; last row/column replicate the previous
    difference; only the centre pixel's validity masks the result.  reference
    structures/rgbdimages.py:710-743."""
    dh = torch.zeros_like(V)
    dv = torch.zeros_like(V)
    dh[..., :-1, :] = V[..., 1:, :] - V[..., :-1, :]
    dv[..., :-1, :, :] = V[..., 1:, :, :] - V[..., :-1, :, :]
    dh[..., -1, :] = dh[..., -2, :]
    dv[..., -1, :, :] = dv[..., -2, :, :]
    n = torch.cross(dh, dv, dim=-1)
    nn = n.norm(dim=-1).unsqueeze(-1)
    n = n / torch.where(nn == 0, torch.ones_like(nn), nn)
    return n * (depth > 0).to(n.dtype)


def global_vertex_map(V: torch.Tensor, depth: torch.Tensor, poses) -> torch.Tensor:
    """reference structures/rgbdimages.py:681-708."""
    if poses is None:
        return V.clone()
    B, L = V.shape[:2]
    R, t = poses[..., :3, :3], poses[..., :3, 3]
    G = torch.einsum("bsjc,bshwc->bshwj", R, V) + t.view(B, L, 1, 1, 3)
    return G * (depth > 0).to(G.dtype)


def global_normal_map(N: torch.Tensor, poses) -> torch.Tensor:
    """reference structures/rgbdimages.py:745-762 (no re-mask)."""
    if poses is None:
        return N.clone()
    return torch.einsum("bsjc,bshwc->bshwj", poses[..., :3, :3], N)


def all_maps(depth, K, poses):
    V = vertex_map(depth, K)
    N = normal_map(V, depth)
    return V, N, global_vertex_map(V, depth, poses), global_normal_map(N, poses)
