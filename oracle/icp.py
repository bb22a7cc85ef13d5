"""Point-to-plane ICP / gradICP restatement (fp32 torch CPU ops + the C nearest-neighbour)."""
from typing import Optional

import torch

from .cloud import Cloud
from .geometry import se3_exp, transform_pointcloud
from .knn import knn1


def solve_linear_system(A: torch.Tensor, b: torch.Tensor, damp) -> torch.Tensor:
    """x = inverse(A^T A + damp I) (A^T b).  reference odometry/icputils.py:78-90."""
    damp = damp if torch.is_tensor(damp) else torch.tensor(damp, dtype=A.dtype)
    At = torch.transpose(A, 0, 1)
    AtA = torch.matmul(At, A) + torch.eye(A.shape[1]) * damp
    return torch.matmul(torch.inverse(AtA), torch.matmul(At, b))


def gauss_newton_solve(src: torch.Tensor, tgt: torch.Tensor, nrm: torch.Tensor, dist_thresh=None):
    """src (1,Ns,3), tgt/nrm (1,Nt,3) -> A (Nf,6), b (Nf,1), idx (Nf,).
    reference odometry/icputils.py:196-232.  NB: the threshold is compared with the SQUARED
    distance (:203-207)."""
    src, tgt, nrm = src.contiguous(), tgt.contiguous(), nrm.contiguous()
    d2, idx = knn1(src[0], tgt[0])
    keep = torch.ones_like(d2, dtype=torch.bool) if dist_thresh is None else d2 < dist_thresh
    idx = idx[keep].long()
    sx = src[0, keep, 0].view(-1, 1)
    sy = src[0, keep, 1].view(-1, 1)
    sz = src[0, keep, 2].view(-1, 1)
    d = torch.index_select(tgt, 1, idx)
    n = torch.index_select(nrm, 1, idx)
    dx, dy, dz = d[0, :, 0].view(-1, 1), d[0, :, 1].view(-1, 1), d[0, :, 2].view(-1, 1)
    nx, ny, nz = n[0, :, 0].view(-1, 1), n[0, :, 1].view(-1, 1), n[0, :, 2].view(-1, 1)
    A = torch.cat([nx, ny, nz, nz * sy - ny * sz, nx * sz - nz * sx, ny * sx - nx * sy], 1)
    b = nx * (dx - sx) + ny * (dy - sy) + nz * (dz - sz)
    return A, b, idx


def point_to_plane_ICP(src, tgt, nrm, T0: Optional[torch.Tensor] = None, numiters: int = 20,
                       damp: float = 1e-8, dist_thresh=None, trace: Optional[list] = None):
    """LM loop with a hard accept/reject.  reference odometry/icputils.py:310-367."""
    src, tgt, nrm = src.contiguous(), tgt.contiguous(), nrm.contiguous()
    damp = torch.tensor(damp, dtype=src.dtype)
    T0 = torch.eye(4, dtype=src.dtype) if T0 is None else T0
    src = transform_pointcloud(src[0], T0).unsqueeze(0)
    T = T0
    idx = None
    for _ in range(numiters):
        A, b, idx = gauss_newton_solve(src, tgt, nrm, dist_thresh)
        r = b[:, 0]
        xi = solve_linear_system(A, b, damp)
        dT = se3_exp(xi)
        err = torch.dot(r.t(), r)
        look = transform_pointcloud(src[0], dT).unsqueeze(0)
        _, b1, _ = gauss_newton_solve(look, tgt, nrm, dist_thresh)
        r1 = b1[:, 0]
        new_err = torch.dot(r1.t(), r1)
        accept = bool(new_err < err)
        if trace is not None:
            trace.append(dict(AtA=A.t() @ A, Atb=A.t() @ b, err=err.clone(), new_err=new_err.clone(),
                              accept=accept, damp=damp.clone(), xi=xi.clone(), idx=idx.clone()))
        if accept:
            src = look
            damp = damp / 2
            T = torch.mm(dT, T)
        else:
            damp = damp * 2
    return T, idx


def point_to_plane_gradICP(src, tgt, nrm, T0=None, numiters: int = 20, damp: float = 1e-8,
                           dist_thresh=None, lambda_max=2.0, B=1.0, B2=1.0, nu=200.0,
                           trace: Optional[list] = None):
    """Smooth (gradLM) variant.  reference odometry/icputils.py:479-545."""
    src, tgt, nrm = src.contiguous(), tgt.contiguous(), nrm.contiguous()
    damp = torch.tensor(damp, dtype=src.dtype)
    lambda_min = 1 / lambda_max
    T0 = torch.eye(4, dtype=src.dtype) if T0 is None else T0
    src = transform_pointcloud(src[0], T0).unsqueeze(0)
    T = T0
    idx = None
    for _ in range(numiters):
        A, b, idx = gauss_newton_solve(src, tgt, nrm, dist_thresh)
        r = b[:, 0]
        xi = solve_linear_system(A, b, damp)
        dT = se3_exp(xi)
        err = torch.dot(r.t(), r)
        look = transform_pointcloud(src[0], dT).unsqueeze(0)
        _, b1, _ = gauss_newton_solve(look, tgt, nrm, dist_thresh)
        r1 = b1[:, 0]
        new_err = torch.dot(r1.t(), r1)
        diff = (new_err - err).clamp(-70.0, 70.0)
        damp_new = lambda_min + (lambda_max - lambda_min) / (1 + torch.exp(-B * diff))
        if trace is not None:
            trace.append(dict(AtA=A.t() @ A, Atb=A.t() @ b, err=err.detach().clone(),
                              new_err=new_err.detach().clone(), damp=damp.detach().clone(),
                              xi=xi.detach().clone(), idx=idx.clone()))
        damp = damp * damp_new
        sig = 1 / ((1 + torch.exp(-B2 * diff)) ** (1 / nu))
        dT = se3_exp(sig * xi)
        src = transform_pointcloud(src[0], dT).unsqueeze(0)
        T = torch.mm(dT, T)
    return T, idx


def downsample_frame(gv, gn, rgb, depth, ds: int) -> Cloud:
    """[::ds, ::ds] subsample + valid-depth mask in row-major order.  Inputs are one-frame
    maps (B,1,H,W,C).  reference odometry/icputils.py:651-669."""
    B = gv.shape[0]
    mask = (depth > 0).squeeze(-1)[..., ::ds, ::ds]
    pts = [gv[b][..., ::ds, ::ds, :][mask[b]] for b in range(B)]
    nrm = [gn[b][..., ::ds, ::ds, :][mask[b]] for b in range(B)]
    col = [rgb[b][..., ::ds, ::ds, :][mask[b]] for b in range(B)]
    return Cloud(pts, nrm, col)


def downsample_map(cloud: Cloud, pc2im: torch.Tensor, ds: int) -> Cloud:
    """Keep rows whose (h, w) lie on the ds-grid, gather map attributes by n, order preserved.
    reference odometry/icputils.py:593-620."""
    B = len(cloud)
    t = pc2im[pc2im[..., 2] % ds == 0]
    t = t[t[..., 3] % ds == 0]
    sel = [t[t[..., 0] == b][..., 1] for b in range(B)]
    g = lambda xs: None if xs is None else [xs[b][sel[b]] for b in range(B)]
    return Cloud(g(cloud.points), g(cloud.normals), g(cloud.colors))


def provide(maps: Cloud, frames: Cloud, odom: str = "icp", numiters=20, damp=1e-8, dist_thresh=None,
            lambda_max=2.0, B=1.0, B2=1.0, nu=200.0) -> torch.Tensor:
    """Per-batch loop of the odometry providers -> (B,1,4,4).  reference odometry/icp.py:80-97,
    odometry/gradicp.py:101-122 (src = frame, tgt = map)."""
    out = []
    for b in range(len(maps)):
        args = (frames.points[b].unsqueeze(0), maps.points[b].unsqueeze(0),
                maps.normals[b].unsqueeze(0), torch.eye(4))
        if odom == "icp":
            T, _ = point_to_plane_ICP(*args, numiters=numiters, damp=damp, dist_thresh=dist_thresh)
        else:
            T, _ = point_to_plane_gradICP(*args, numiters=numiters, damp=damp, dist_thresh=dist_thresh,
                                          lambda_max=lambda_max, B=B, B2=B2, nu=nu)
        out.append(T)
    return torch.stack(out).unsqueeze(1)
