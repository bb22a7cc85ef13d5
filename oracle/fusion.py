"""PointFusion map update restatement (fp32 torch CPU ops).  A live frame is passed as a dict
``fr`` with one-frame maps: depth (B,1,H,W,1), rgb, V, N, gV, gN (B,1,H,W,3), K (B,1,4,4),
pose (B,1,4,4)."""
import warnings

import torch

from .cloud import Cloud
from .geometry import inverse_transformation, project_points
from .maps import all_maps


def make_frame(rgb, depth, K, pose) -> dict:
    V, N, gV, gN = all_maps(depth, K, pose)
    return dict(rgb=rgb, depth=depth, K=K, pose=pose, V=V, N=N, gV=gV, gN=gN)


def get_alpha(points: torch.Tensor, sigma, dim: int = -1, keepdim: bool = False, eps: float = 1e-7):
    """reference slam/fusionutils.py:69-73."""
    a = torch.exp(-torch.sum(points ** 2, dim, keepdim=keepdim) / (2 * (sigma ** 2)))
    return torch.clamp(a, min=eps, max=1.01)


def find_active_map_points(cloud: Cloud, fr: dict) -> torch.Tensor:
    """Rows [b, n, h, w] (int64) of map points that project inside the frame, in (b, n) order.
    reference slam/fusionutils.py:235-287, structures/pointclouds.py:399-430,466-614."""
    if not cloud.has_points:
        return torch.empty((0, 4), dtype=torch.int64)
    H, W = fr["depth"].shape[2:4]
    Tinv = inverse_transformation(fr["pose"].squeeze(1))
    pts = cloud.padded("points")
    nonpad = cloud.nonpad_mask()
    npf = nonpad.to(pts.dtype).unsqueeze(-1)
    # transform = rotate (einsum with R^T) then offset (masked by nonpad)
    Rt = Tinv[..., :3, :3].transpose(-1, -2)
    p = torch.einsum("bij,bjk->bik", pts, Rt)
    p = p + Tinv[..., :3, 3].unsqueeze(-2) * npf
    front = p[..., -1] > 0
    uv = project_points(p, fr["K"].squeeze(1))
    uv1 = torch.nn.functional.pad(uv, (0, 1), "constant", 1.0) * npf
    img = uv1[..., :-1]
    inside = ((img[..., 0] > -1e-3) & (img[..., 0] < W - 0.999) & (img[..., 1] > -1e-3)
              & (img[..., 1] < H - 0.999) & front & nonpad)
    pos = img.round().long()
    hw = torch.cat([pos[..., 1:2].clamp(0, H - 1), pos[..., 0:1].clamp(0, W - 1)], -1)
    Bn, Nn = hw.shape[:2]
    bb, nn = torch.meshgrid([torch.arange(Bn), torch.arange(Nn)], indexing="ij")
    table = torch.cat([bb.unsqueeze(-1), nn.unsqueeze(-1), hw], -1)[inside]
    if table.shape[0] == 0:
        warnings.warn("No active map points were found")
    return table


def find_similar_map_points(cloud: Cloud, fr: dict, pc2im: torch.Tensor, dist_th, dot_th):
    """reference slam/fusionutils.py:363-411 (Euclidean distance < dist_th, normal dot > dot_th)."""
    if not cloud.has_points or pc2im.shape[0] == 0:
        return torch.empty((0, 4), dtype=torch.int64), torch.empty(0, dtype=torch.bool)
    mp, mn = cloud.padded("points"), cloud.padded("normals")
    b, n, h, w = pc2im[:, 0], pc2im[:, 1], pc2im[:, 2], pc2im[:, 3]
    fp = torch.zeros_like(mp)
    fn = torch.zeros_like(mn)
    fp[b, n] = fr["gV"][b, 0, h, w]
    fn[b, n] = fr["gN"][b, 0, h, w]
    close = (fp - mp).norm(dim=-1) < dist_th
    dots = (fn * mn).sum(-1)
    if dots.max() > 1.001:
        warnings.warn("Max of dot product was {0} > 1. Inputs were not normalized along dim ({1}). "
                      "Was this intentional?".format(dots.max(), -1), RuntimeWarning)
    mask = (close & (dots > dot_th))[b, n]
    out = pc2im[mask]
    if len(out) == 0:
        warnings.warn("No similar map points were found (despite total {0} active points across "
                      "the batch)".format(pc2im.shape[0]), RuntimeWarning)
    return out, mask


def find_best_unique_correspondences(cloud: Cloud, fr: dict, pc2im: torch.Tensor) -> torch.Tensor:
    """Per (b,h,w) keep the candidate minimising (1/(ccount+1e-20), squared ray distance, n), all
    compared as fp32; output sorted by (b,h,w).  reference slam/fusionutils.py:473-546 (a
    lexicographic row sort via torch.unique(dim=0))."""
    if not cloud.has_points or pc2im.shape[0] == 0:
        return torch.empty((0, 4), dtype=torch.int64)
    b, n, h, w = pc2im[:, 0], pc2im[:, 1], pc2im[:, 2], pc2im[:, 3]
    inv_c = 1 / (cloud.padded("feats")[b, n] + 1e-20)
    ray = ((cloud.padded("points")[b, n] - fr["gV"][b, 0, h, w]) ** 2).sum(-1).unsqueeze(1)
    crit = torch.cat([pc2im[:, 0:1].float(), pc2im[:, 2:4].float(), inv_c, ray,
                      pc2im[:, 1:2].float()], -1)
    srt = torch.unique(crit.detach(), dim=0)
    first = torch.ones(srt.shape[0], dtype=torch.bool)
    first[1:] = (srt[1:, :3] - srt[:-1, :3] != 0).any(-1)
    u = srt[first]
    return torch.cat([u[:, 0:1].long(), u[:, -1:].long(), u[:, 1:3].long()], -1)


def find_correspondences(cloud: Cloud, fr: dict, dist_th, dot_th) -> torch.Tensor:
    """reference slam/fusionutils.py:572-577."""
    t = find_active_map_points(cloud, fr)
    t, _ = find_similar_map_points(cloud, fr, t, dist_th, dot_th)
    return find_best_unique_correspondences(cloud, fr, t)


def fuse_with_map(cloud: Cloud, fr: dict, pc2im: torch.Tensor, sigma) -> Cloud:
    """Confidence-weighted merge of matched points, then append of unmatched valid pixels in
    (h,w) row-major order.  reference slam/fusionutils.py:654-722.  Functional: returns a new
    Cloud."""
    gV, gN, rgb = fr["gV"], fr["gN"], fr["rgb"]
    alpha = get_alpha(fr["V"], dim=4, keepdim=True, sigma=sigma)
    merged = cloud
    if cloud.has_points and pc2im.shape[0] != 0:
        b, n, h, w = pc2im[:, 0], pc2im[:, 1], pc2im[:, 2], pc2im[:, 3]
        mp, mn, mc, cc = (cloud.padded(k) for k in ("points", "normals", "colors", "feats"))
        fp, fn, fc, fa = (torch.zeros_like(x) for x in (mp, mn, mc, cc))
        fp[b, n] = gV[b, 0, h, w]
        fn[b, n] = gN[b, 0, h, w]
        fc[b, n] = rgb[b, 0, h, w]
        fa[b, n] = alpha[b, 0, h, w]
        cc2 = cc + fa
        inv = 1 / torch.where(cc2 == 0, torch.ones_like(cc2), cc2)
        merged = Cloud(cloud.points, cloud.normals, cloud.colors, cloud.feats)
        merged.set_from_padded("points", ((cc * mp) + (fa * fp)) * inv)
        merged.set_from_padded("normals", ((cc * mn) + (fa * fn)) * inv)
        merged.set_from_padded("colors", ((cc * mc) + (fa * fc)) * inv)
        merged.set_from_padded("feats", cc2)
    new = torch.ones_like(gV[..., 0], dtype=bool)
    if cloud.has_points and pc2im.shape[0] != 0:
        new[pc2im[:, 0], 0, pc2im[:, 2], pc2im[:, 3]] = 0
    new = new * (fr["depth"] > 0).squeeze(-1)
    B = new.shape[0]
    fresh = Cloud([gV[b][new[b]] for b in range(B)], [gN[b][new[b]] for b in range(B)],
                  [rgb[b][new[b]] for b in range(B)], [alpha[b][new[b]] for b in range(B)])
    return merged.append(fresh)


def update_map_fusion(cloud: Cloud, fr: dict, dist_th, dot_th, sigma) -> Cloud:
    """reference slam/fusionutils.py:785-789."""
    return fuse_with_map(cloud, fr, find_correspondences(cloud, fr, dist_th, dot_th), sigma)


def update_map_aggregate(cloud: Cloud, fr: dict) -> Cloud:
    """Append every valid pixel unmerged, no confidence counts.  reference
    slam/fusionutils.py:754-758, structures/utils.py:38-57."""
    B = fr["gV"].shape[0]
    m = (fr["depth"] > 0).squeeze(-1)
    fresh = Cloud([fr["gV"][b][m[b]] for b in range(B)], [fr["gN"][b][m[b]] for b in range(B)],
                  [fr["rgb"][b][m[b]] for b in range(B)])
    return cloud.append(fresh)
