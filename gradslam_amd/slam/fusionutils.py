"""PointFusion map update on the MI355X (reference slam/fusionutils.py).

Same free functions, arguments, warnings and error contracts as the reference.  The arithmetic runs
in HIP kernels (gradslam_amd/csrc/project.hip, fusion.hip, maps.hip): fused transform + project +
in-frame test + stable compaction for the active-point search; gather + threshold for the
similarity test; per-pixel atomic-min keys instead of a lexicographic row sort for the unique best
correspondence; one pass over the map for the confidence-weighted merge; stable compaction for the
appended points.
"""
import warnings
from typing import Union

import torch

from .. import ops
from ..structures.pointclouds import Pointclouds
from ..structures.rgbdimages import RGBDImages
from ..structures.utils import pointclouds_from_rgbdimages

__all__ = ["update_map_fusion", "update_map_aggregate"]


def get_alpha(points: torch.Tensor, sigma: Union[torch.Tensor, float, int], dim: int = -1, keepdim: bool = False,
              eps: float = 1e-7) -> torch.Tensor:
    """Sample confidence alpha = clamp(exp(-|p|^2 / (2 sigma^2)), eps, 1.01) (reference :16-73)."""
    if not torch.is_tensor(points):
        raise TypeError("Expected input points to be of type torch.Tensor. Got {0} instead.".format(type(points)))
    if not (torch.is_tensor(sigma) or isinstance(sigma, (float, int))):
        raise TypeError("Expected input sigma to be of type torch.Tensor or float or int. Got {0} instead.".format(type(sigma)))
    if not isinstance(eps, float):
        raise TypeError("Expected input eps to be of type float. Got {0} instead.".format(type(eps)))
    if points.shape[dim] != 3:
        raise ValueError("Expected length of dim-th ({0}th) dimension to be 3. Got {1} instead.".format(dim, points.shape[dim]))
    if torch.is_tensor(sigma) and sigma.ndim != 0:
        raise ValueError("Expected sigma.ndim to be 0 (scalar). Got {0}.".format(sigma.ndim))
    ops.require_hip(points, op="get_alpha")
    sig = float(sigma)  # a 0-dim tensor sigma is read once
    alpha = ops.get_alpha_lastdim(points.movedim(dim, -1), sig, eps)
    return alpha.unsqueeze(dim) if keepdim else alpha


def _check_pair(tensor1, tensor2, th, th_name):
    if not torch.is_tensor(tensor1):
        raise TypeError("Expected input tensor1 to be of type torch.Tensor. Got {0} instead.".format(type(tensor1)))
    if not torch.is_tensor(tensor2):
        raise TypeError("Expected input tensor2 to be of type torch.Tensor. Got {0} instead.".format(type(tensor2)))
    if not isinstance(th, (float, int)):
        raise TypeError("Expected input {0} to be of type float or int. Got {1} instead.".format(th_name, type(th)))
    if tensor1.shape != tensor2.shape:
        raise ValueError("tensor1 and tensor2 should have the same shape, but had shapes {0} and {1} respectively.".format(
            tensor1.shape, tensor2.shape))


def are_points_close(tensor1: torch.Tensor, tensor2: torch.Tensor, dist_th: Union[float, int], dim: int = -1):
    """|t1 - t2|_2 < dist_th along `dim` (reference :76-130).  Thin API helper; the fused similarity
    kernel applies the same test inside `find_similar_map_points`."""
    _check_pair(tensor1, tensor2, dist_th, "dist_th")
    if tensor1.shape[dim] != 3:
        raise ValueError("Expected length of input tensors' dim-th ({0}th) dimension to be 3. Got {1} instead.".format(
            dim, tensor1.shape[dim]))
    return (tensor1 - tensor2).norm(dim=dim) < dist_th


def are_normals_similar(tensor1: torch.Tensor, tensor2: torch.Tensor, dot_th: Union[float, int], dim: int = -1):
    """t1 . t2 > dot_th along `dim`, warning when inputs are not normalised (reference :133-195)."""
    _check_pair(tensor1, tensor2, dot_th, "dot_th")
    if tensor1.shape[dim] != 3:
        raise ValueError("Expected length of input tensors' dim-th ({0}th) dimension to be 3. Got {1} instead.".format(
            dim, tensor1.shape[dim]))
    dot_res = (tensor1 * tensor2).sum(dim)
    if dot_res.max() > 1.001:
        warnings.warn("Max of dot product was {0} > 1. Inputs were not normalized along dim ({1}). Was this "
                      "intentional?".format(dot_res.max(), dim), RuntimeWarning)
    return dot_res > dot_th


def _check_pc_rgbd(pointclouds, rgbdimages):
    if not isinstance(pointclouds, Pointclouds):
        raise TypeError("Expected pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(type(pointclouds)))
    if not isinstance(rgbdimages, RGBDImages):
        raise TypeError("Expected rgbdimages to be of type gradslam.RGBDImages. Got {0}.".format(type(rgbdimages)))


def _check_table(pc2im_bnhw):
    if not torch.is_tensor(pc2im_bnhw):
        raise TypeError("Expected input pc2im_bnhw to be of type torch.Tensor. Got {0} instead.".format(type(pc2im_bnhw)))
    if pc2im_bnhw.dtype != torch.int64:
        raise TypeError("Expected input pc2im_bnhw to have dtype of torch.int64 (torch.long), not {0}.".format(pc2im_bnhw.dtype))


def _check_table_shape(pc2im_bnhw):
    if pc2im_bnhw.ndim != 2:
        raise ValueError("Expected pc2im_bnhw.ndim of 2. Got {0}.".format(pc2im_bnhw.ndim))
    if pc2im_bnhw.shape[1] != 4:
        raise ValueError("Expected pc2im_bnhw.shape[1] to be 4. Got {0}.".format(pc2im_bnhw.shape[1]))


def _check_batch(pointclouds, rgbdimages):
    if len(rgbdimages) != len(pointclouds):
        raise ValueError("Expected equal batch sizes for pointclouds and rgbdimages. Got {0} and {1} respectively.".format(
            len(pointclouds), len(rgbdimages)))


def _cl(rgbdimages: RGBDImages) -> RGBDImages:
    return rgbdimages.to_channels_last() if rgbdimages.channels_first else rgbdimages


def _project(pointclouds: Pointclouds, rgbdimages: RGBDImages, ds: int = 0):
    """(rows buffer, device count) of the active-point search; no host sync."""
    _, _, H, W = rgbdimages.shape
    return ops.project_active_raw(pointclouds.points_padded.detach(), pointclouds._counts_i32(),
                                  rgbdimages.poses.squeeze(1).detach(), rgbdimages.intrinsics.squeeze(1).detach(), H, W, ds)


def find_active_map_points(pointclouds: Pointclouds, rgbdimages: RGBDImages) -> torch.Tensor:
    """Rows [b, n, h, w] of the map points that project inside the live frame, in (b, n) order
    (reference :198-287)."""
    _check_pc_rgbd(pointclouds, rgbdimages)
    if rgbdimages.shape[1] != 1:
        raise ValueError("Expected rgbdimages to have sequence length of 1. Got {0}.".format(rgbdimages.shape[1]))
    device = pointclouds.device
    if not pointclouds.has_points:
        return torch.empty((0, 4), dtype=torch.int64, device=device)
    _check_batch(pointclouds, rgbdimages)
    rows, cnt = _project(pointclouds, rgbdimages)
    pc2im_bnhw = rows[: int(cnt.item())]
    if pc2im_bnhw.shape[0] == 0:
        warnings.warn("No active map points were found")
    return pc2im_bnhw


def find_similar_map_points(pointclouds: Pointclouds, rgbdimages: RGBDImages, pc2im_bnhw: torch.Tensor,
                            dist_th: Union[float, int], dot_th: Union[float, int]):
    """Rows whose map point is within `dist_th` of, and has a normal within `dot_th` of, the live
    frame pixel it projects to; also the bool mask over the input rows (reference :290-411)."""
    _check_pc_rgbd(pointclouds, rgbdimages)
    _check_table(pc2im_bnhw)
    if rgbdimages.shape[1] != 1:
        raise ValueError("Expected rgbdimages to have sequence length of 1. Got {0}.".format(rgbdimages.shape[1]))
    _check_table_shape(pc2im_bnhw)
    device = pointclouds.device
    if not pointclouds.has_points or pc2im_bnhw.shape[0] == 0:
        return (torch.empty((0, 4), dtype=torch.int64, device=device), torch.empty(0, dtype=torch.bool, device=device))
    _check_batch(pointclouds, rgbdimages)
    if not pointclouds.has_normals:
        raise ValueError("Pointclouds must have normals for finding similar map points, but did not.")
    rgbdimages = _cl(rgbdimages)
    rows = pc2im_bnhw.contiguous()
    P = rows.shape[0]
    keep, max_dot = ops.fusion_similar_raw(rows, ops.dev_int(P, rows.device), P, rgbdimages.global_vertex_map.detach(),
                                           rgbdimages.global_normal_map.detach(), pointclouds.points_padded.detach(),
                                           pointclouds.normals_padded.detach(), dist_th, dot_th)
    is_similar_mask = keep[:P].view(torch.bool)
    md = float(max_dot.item())
    if md > 1.001:
        warnings.warn("Max of dot product was {0} > 1. Inputs were not normalized along dim ({1}). Was this "
                      "intentional?".format(md, -1), RuntimeWarning)
    pc2im_bnhw_similar = ops.compact_rows(rows, keep[:P])
    if len(pc2im_bnhw_similar) == 0:
        warnings.warn("No similar map points were found (despite total {0} active points across the batch)".format(
            pc2im_bnhw.shape[0]), RuntimeWarning)
    return pc2im_bnhw_similar, is_similar_mask


def find_best_unique_correspondences(pointclouds: Pointclouds, rgbdimages: RGBDImages, pc2im_bnhw: torch.Tensor):
    """Among rows sharing a live-frame pixel keep the one with the highest confidence count, then
    the smallest ray distance, then the smallest n; rows come back sorted by (b, h, w)
    (reference :414-546)."""
    if not isinstance(pointclouds, Pointclouds):
        raise TypeError("Expected pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(type(pointclouds)))
    _check_table(pc2im_bnhw)
    if rgbdimages.shape[1] != 1:
        raise ValueError("Expected rgbdimages to have sequence length of 1. Got {0}.".format(rgbdimages.shape[1]))
    _check_table_shape(pc2im_bnhw)
    device = pointclouds.device
    if not pointclouds.has_points or pc2im_bnhw.shape[0] == 0:
        return torch.empty((0, 4), dtype=torch.int64, device=device)
    _check_batch(pointclouds, rgbdimages)
    if not pointclouds.has_features:
        raise ValueError("Pointclouds must have features for finding best unique correspondences, but did not.")
    rgbdimages = _cl(rgbdimages)
    rows = pc2im_bnhw.contiguous()
    P = rows.shape[0]
    out, cnt = ops.fusion_unique_raw(rows, None, ops.dev_int(P, rows.device), P, rgbdimages.global_vertex_map.detach(),
                                     pointclouds.points_padded.detach(), pointclouds.features_padded.detach())
    return out[: int(cnt.item())]


def find_correspondences(pointclouds: Pointclouds, rgbdimages: RGBDImages, dist_th: Union[float, int],
                         dot_th: Union[float, int]) -> torch.Tensor:
    """active -> similar -> best unique (reference :549-577), chained on the device: the
    intermediate tables are never compacted or copied to the host; one sync returns the three
    counts the warnings and the output shape need."""
    _check_pc_rgbd(pointclouds, rgbdimages)
    if rgbdimages.shape[1] != 1:
        raise ValueError("Expected rgbdimages to have sequence length of 1. Got {0}.".format(rgbdimages.shape[1]))
    device = pointclouds.device
    if not pointclouds.has_points:
        return torch.empty((0, 4), dtype=torch.int64, device=device)
    _check_batch(pointclouds, rgbdimages)
    if not pointclouds.has_normals:
        raise ValueError("Pointclouds must have normals for finding similar map points, but did not.")
    if not pointclouds.has_features:
        raise ValueError("Pointclouds must have features for finding best unique correspondences, but did not.")
    rgbdimages = _cl(rgbdimages)
    rows, cnt = _project(pointclouds, rgbdimages)
    cap = rows.shape[0]
    gV, gN = rgbdimages.global_vertex_map.detach(), rgbdimages.global_normal_map.detach()
    mp = pointclouds.points_padded.detach()
    keep, max_dot = ops.fusion_similar_raw(rows, cnt, cap, gV, gN, mp, pointclouds.normals_padded.detach(), dist_th, dot_th)
    out, ucnt = ops.fusion_unique_raw(rows, keep, cnt, cap, gV, mp, pointclouds.features_padded.detach())
    stats = torch.stack([cnt[0].float(), keep.sum(dtype=torch.float32), ucnt[0].float(), max_dot[0]]).tolist()
    n_active, n_similar, n_unique, md = int(stats[0]), int(stats[1]), int(stats[2]), stats[3]
    if n_active == 0:
        warnings.warn("No active map points were found")
        return torch.empty((0, 4), dtype=torch.int64, device=device)
    if md > 1.001:
        warnings.warn("Max of dot product was {0} > 1. Inputs were not normalized along dim ({1}). Was this "
                      "intentional?".format(md, -1), RuntimeWarning)
    if n_similar == 0:
        warnings.warn("No similar map points were found (despite total {0} active points across the batch)".format(
            n_active), RuntimeWarning)
    return out[:n_unique]


def fuse_with_map(pointclouds: Pointclouds, rgbdimages: RGBDImages, pc2im_bnhw: torch.Tensor,
                  sigma: Union[torch.Tensor, float, int], inplace: bool = False) -> Pointclouds:
    """Merge the matched live-frame points into the map (confidence-weighted running average) and
    append the unmatched valid pixels in (h, w) row-major order (reference :580-722)."""
    _check_pc_rgbd(pointclouds, rgbdimages)
    _check_table(pc2im_bnhw)
    _check_table_shape(pc2im_bnhw)
    if pointclouds.has_points:
        if not pointclouds.has_normals:
            raise ValueError("Pointclouds must have normals for map fusion, but did not.")
        if not pointclouds.has_colors:
            raise ValueError("Pointclouds must have colors for map fusion, but did not.")
        if not pointclouds.has_features:
            raise ValueError("Pointclouds must have features (ccounts) for map fusion, but did not.")
    ops.require_hip(pc2im_bnhw, op="fuse_with_map")
    rgbdimages = _cl(rgbdimages)
    vertex_maps, normal_maps = rgbdimages.global_vertex_map, rgbdimages.global_normal_map
    rgb_image = rgbdimages.rgb_image
    alpha_image = get_alpha(rgbdimages.vertex_map, dim=4, keepdim=True, sigma=sigma)
    B, _, H, W = rgbdimages.shape
    rows = pc2im_bnhw.contiguous()
    U = rows.shape[0]
    has_match = pointclouds.has_points and U != 0

    if has_match:
        new_p, new_n, new_c, new_f = ops.fusion_merge(rows, pointclouds._counts_i32(), vertex_maps, normal_maps, rgb_image,
                                                      alpha_image, pointclouds.points_padded, pointclouds.normals_padded,
                                                      pointclouds.colors_padded, pointclouds.features_padded)
        # like the reference, the merge lands in the object that was passed in, inplace or not
        pointclouds._adopt_padded(new_p, new_n, new_c, new_f)

    mask = ops.fusion_new_mask_raw(rgbdimages.depth_image, rows if has_match else None,
                                   ops.dev_int(U, rows.device) if has_match else None, U if has_match else 0)
    grad_mode = torch.is_grad_enabled() and any(t.requires_grad for t in (vertex_maps, normal_maps, rgb_image, alpha_image))
    flat = lambda x, b, c: x[b].reshape(-1, c)
    if grad_mode:
        per_b = [ops.mask_select_multi([flat(vertex_maps, b, 3), flat(normal_maps, b, 3), flat(rgb_image, b, 3),
                                        flat(alpha_image, b, 1)], mask[b].reshape(-1)) for b in range(B)]
        new = [[per_b[b][a] for b in range(B)] for a in range(4)]
    else:
        bufs, cnts = [], []
        for b in range(B):
            outs, c = ops.compact_multi_raw([flat(vertex_maps, b, 3), flat(normal_maps, b, 3), flat(rgb_image, b, 3),
                                             flat(alpha_image, b, 1)], mask[b].reshape(-1))
            bufs.append(outs)
            cnts.append(c)
        n_new = torch.cat(cnts).tolist()  # the one host sync of the map update
        new = [[bufs[b][a][: n_new[b]] for b in range(B)] for a in range(4)]
    new_pointclouds = Pointclouds(points=new[0], normals=new[1], colors=new[2], features=new[3])
    if not inplace:
        pointclouds = pointclouds.clone()
    pointclouds.append_points(new_pointclouds)
    return pointclouds


def update_map_aggregate(pointclouds: Pointclouds, rgbdimages: RGBDImages, inplace: bool = False) -> Pointclouds:
    """Append every valid live-frame point to the map, unmerged (reference :725-758)."""
    _check_pc_rgbd(pointclouds, rgbdimages)
    new_pointclouds = pointclouds_from_rgbdimages(rgbdimages, global_coordinates=True)
    if not inplace:
        pointclouds = pointclouds.clone()
    pointclouds.append_points(new_pointclouds)
    return pointclouds


def update_map_fusion(pointclouds: Pointclouds, rgbdimages: RGBDImages, dist_th: Union[float, int],
                      dot_th: Union[float, int], sigma: Union[torch.Tensor, float, int], inplace: bool = False) -> Pointclouds:
    """PointFusion update: correspondences, then merge + append (reference :761-789)."""
    pc2im_bnhw = find_correspondences(pointclouds, rgbdimages, dist_th, dot_th)
    return fuse_with_map(pointclouds, rgbdimages, pc2im_bnhw, sigma, inplace)
