"""Restatement of the few geometry helpers the hot path calls (fp32 torch CPU ops)."""
import torch

SMALL_ANGLE = 1e-6  # reference geometry/se3utils.py:8


def inverse_intrinsics(K: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """reference geometry/projutils.py:437-450 (note the +eps on the focal lengths)."""
    Kinv = torch.zeros_like(K)
    fx, fy = K[..., 0, 0], K[..., 1, 1]
    cx, cy = K[..., 0, 2], K[..., 1, 2]
    Kinv[..., 0, 0] = 1.0 / (fx + eps)
    Kinv[..., 1, 1] = 1.0 / (fy + eps)
    Kinv[..., 0, 2] = -1.0 * cx / (fx + eps)
    Kinv[..., 1, 2] = -1.0 * cy / (fy + eps)
    Kinv[..., 2, 2] = 1
    Kinv[..., -1, -1] = 1
    return Kinv


def pixel_grid(B: int, L: int, H: int, W: int) -> torch.Tensor:
    """(w, h, 1) per pixel: reference structures/rgbdimages.py:647-661 with the float32
    linspace meshgrid of geometry/geometryutils.py:599-608."""
    hs = torch.linspace(0, H - 1, H)
    ws = torch.linspace(0, W - 1, W)
    hh, ww = torch.meshgrid([hs, ws], indexing="ij")
    pix = torch.stack([ww, hh, torch.ones_like(ww)], -1)
    return pix.view(1, 1, H, W, 3).repeat(B, L, 1, 1, 1)


def so3_hat(w: torch.Tensor) -> torch.Tensor:
    """reference geometry/se3utils.py:11-26."""
    m = torch.zeros(3, 3).type(w.dtype).to(w.device)
    m[0, 1], m[1, 0] = -w[2], w[2]
    m[0, 2], m[2, 0] = w[1], -w[1]
    m[1, 2], m[2, 1] = -w[0], w[0]
    return m


def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    """reference geometry/se3utils.py:77-115.  NB the small-angle branch uses V = I + w^ (sic)."""
    v, w = xi[:3], xi[3:]
    what = so3_hat(w)
    eye = torch.eye(3, 3).type(w.dtype).to(w.device)
    if w.norm() < SMALL_ANGLE:
        R = eye + what
        V = eye + what
    else:
        th = w.norm()
        s, c = th.sin(), th.cos()
        what2 = what.mm(what)
        A = s / th
        Bc = (1 - c) / torch.pow(th, 2)
        C = (th - s) / torch.pow(th, 3)
        R = eye + A * what + Bc * what2
        V = eye + Bc * what + C * what2
    t = torch.mm(V, v.view(3, 1))
    last = torch.tensor([0, 0, 0, 1]).type(w.dtype).to(w.device)
    return torch.cat((torch.cat((R, t), dim=1), last.unsqueeze(0)), dim=0)


def transform_pointcloud(pc: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """reference geometry/geometryutils.py:780-792: (R @ pc^T + t)^T."""
    R, t = T[:3, :3], T[:3, 3]
    return torch.transpose(torch.matmul(R, torch.transpose(pc, 0, 1)) + t.unsqueeze(1), 0, 1)


def inverse_transformation(T: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.linalg.inverse_transformation semantics (R^T, -R^T t); call site
    reference slam/fusionutils.py:249.  Corroborated by geometry/geometryutils.py:205-241."""
    R = T[..., :3, :3]
    t = T[..., :3, 3:4]
    Rt = R.transpose(-1, -2)
    out = torch.zeros_like(T)
    out[..., :3, :3] = out[..., :3, :3] + Rt
    out[..., :3, 3:4] = out[..., :3, 3:4] + torch.matmul(-Rt, t)
    out[..., 3, 3] = out[..., 3, 3] + 1.0
    return out


def compose_transformations(T01: torch.Tensor, T12: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.linalg.compose_transformations semantics; call site reference
    slam/icpslam.py:245-247.  Corroborated by geometry/geometryutils.py:244-301."""
    R = torch.matmul(T01[..., :3, :3], T12[..., :3, :3])
    t = torch.matmul(T01[..., :3, :3], T12[..., :3, 3:4]) + T01[..., :3, 3:4]
    out = torch.zeros_like(T01)
    out[..., :3, :3] = out[..., :3, :3] + R
    out[..., :3, 3:4] = out[..., :3, 3:4] + t
    out[..., 3, 3] = out[..., 3, 3] + 1.0
    return out


def project_points(pts: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """Pinhole projection of (B,N,3) points with (B,4,4) intrinsics; reference
    geometry/projutils.py:206-238 (homogenise, 4x4 matmul, divide by z where z != 0)."""
    ph = torch.nn.functional.pad(pts, (0, 1), "constant", 1.0)
    q = torch.matmul(K.unsqueeze(-3), ph.unsqueeze(-1)).squeeze(-1)
    x, y, z = q[..., 0], q[..., 1], q[..., 2]
    zs = torch.where(z != 0, z, torch.ones_like(z))
    return torch.stack((x / zs, y / zs), dim=-1)
