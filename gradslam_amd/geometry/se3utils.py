"""SE(3) exponential map (reference geometry/se3utils.py:11-26, :77-115).  O(1) algebra kept in
torch so that autograd can differentiate the gradICP path; the no-grad ICP loop uses the device
version inside icp.hip."""
import torch

_eps = 1e-6

__all__ = ["so3_hat", "se3_hat", "so3_exp", "se3_exp"]


def so3_hat(omega: torch.Tensor) -> torch.Tensor:
    assert torch.is_tensor(omega), "Input must be of type torch.tensor."
    w = omega.reshape(-1)
    z = torch.zeros((), dtype=w.dtype, device=w.device)
    return torch.stack([torch.stack([z, -w[2], w[1]]), torch.stack([w[2], z, -w[0]]), torch.stack([-w[1], w[0], z])])


def se3_hat(xi: torch.Tensor) -> torch.Tensor:
    assert torch.is_tensor(xi), "Input must be of type torch.tensor."
    x = xi.reshape(-1)
    out = torch.zeros(4, 4, dtype=x.dtype, device=x.device)
    out[0:3, 0:3] = so3_hat(x[3:])
    out[0:3, 3] = x[:3]
    return out


def _rodrigues(omega: torch.Tensor):
    """(R, V).  NB the small-angle branch uses V = I + w^ exactly like the reference (sic)."""
    what = so3_hat(omega)
    eye = torch.eye(3, dtype=what.dtype, device=what.device)
    th = omega.norm()
    if th < _eps:
        return eye + what, eye + what
    s, c = th.sin(), th.cos()
    what2 = what.mm(what)
    A, B, C = s / th, (1 - c) / torch.pow(th, 2), (th - s) / torch.pow(th, 3)
    return eye + A * what + B * what2, eye + B * what + C * what2


def so3_exp(omega: torch.Tensor) -> torch.Tensor:
    assert torch.is_tensor(omega), "Input must be of type torch.Tensor."
    return _rodrigues(omega.reshape(-1))[0]


def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    """xi = [v ; omega] (6,) or (6,1) -> (4,4)."""
    assert torch.is_tensor(xi), "Input must be of type torch.tensor."
    x = xi.reshape(-1)
    R, V = _rodrigues(x[3:])
    t = torch.mm(V, x[:3].view(3, 1))
    last = torch.tensor([0, 0, 0, 1], dtype=x.dtype, device=x.device).unsqueeze(0)
    return torch.cat((torch.cat((R, t), dim=1), last), dim=0)
