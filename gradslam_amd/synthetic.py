"""Deterministic synthetic RGB-D sequences (TUM-shape) for parity tests and bench.py.

There is no dataset access in the build/bench environment, so every measurement uses this
generator.  The scene is a static "wavy wall" height field in world coordinates,

    z_w = 2.0 + 0.3 * sin(2 x_w) * cos(2 y_w),

ray-cast (fixed-point iteration in float64, exact to fp32) from a pinhole camera that moves
along  pose_s = translate(0.01 s, 0, 0) . rotY(0.002 s),  so consecutive frames are
geometrically consistent views of one surface and ICP has a true answer to recover.
Depth gets an 8-column invalid (zero) band plus a seeded 5 % Bernoulli drop-out, like real
sensors (the reference's own fixture has 11.8 % zeros).  Intrinsics are TUM's
(fx=fy=525, cx=319.5, cy=239.5 at 640x480; reference datasets/tum.py:338-340) scaled to the
requested size.  Colours are uniform random in [0,255).
"""
import numpy as np
import torch

__all__ = ["make_intrinsics", "make_poses", "make_sequence", "make_sequence_cached"]


def make_intrinsics(height: int, width: int) -> torch.Tensor:
    sx, sy = width / 640.0, height / 480.0
    K = np.eye(4, dtype=np.float64)
    K[0, 0], K[1, 1] = 525.0 * sx, 525.0 * sy
    K[0, 2], K[1, 2] = 319.5 * sx, 239.5 * sy
    return torch.from_numpy(K.astype(np.float32)).view(1, 1, 4, 4)


def make_poses(seq_len: int, step_t: float = 0.01, step_r: float = 0.002, phase: float = 0.0) -> np.ndarray:
    out = np.zeros((seq_len, 4, 4), dtype=np.float64)
    for s in range(seq_len):
        a = step_r * s
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        T = np.eye(4)
        T[:3, :3] = R
        T[0, 3] = step_t * s + phase
        out[s] = T
    return out


def _wall(x, y):
    return 2.0 + 0.3 * np.sin(2.0 * x) * np.cos(2.0 * y)


def _raycast(K, pose, height, width):
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    v, u = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)  # z-normalised ray
    d_w = d_cam @ pose[:3, :3].T
    c = pose[:3, 3]
    t = np.full_like(u, 2.0)
    for _ in range(60):
        p = c + t[..., None] * d_w
        t = (_wall(p[..., 0], p[..., 1]) - c[2]) / d_w[..., 2]
    return t  # = camera-frame z because d_cam has unit z


def make_sequence(batch: int = 1, seq_len: int = 2, height: int = 480, width: int = 640, seed: int = 0,
                  dropout: float = 0.05, band: int = 8, step_t: float = 0.01, step_r: float = 0.002):
    """Returns (colors (B,L,H,W,3), depths (B,L,H,W,1), intrinsics (B,1,4,4), poses (B,L,4,4)),
    fp32 CPU tensors.  Batch element b uses seed+b and a lateral phase of 0.05*b."""
    K = make_intrinsics(height, width)
    Kd = K[0, 0].double().numpy()
    colors, depths, poses = [], [], []
    for b in range(batch):
        rng = np.random.RandomState(seed + b)
        P = make_poses(seq_len, step_t, step_r, phase=0.05 * b)
        dl = []
        for s in range(seq_len):
            z = _raycast(Kd, P[s], height, width)
            keep = rng.rand(height, width) >= dropout
            z = z * keep
            if band > 0:
                z[:, width // 3: width // 3 + band] = 0.0
            dl.append(z.astype(np.float32))
        depths.append(np.stack(dl)[..., None])
        colors.append((rng.rand(seq_len, height, width, 3) * 255.0).astype(np.float32))
        poses.append(P.astype(np.float32))
    return (torch.from_numpy(np.stack(colors)), torch.from_numpy(np.stack(depths)),
            K.repeat(batch, 1, 1, 1).contiguous(), torch.from_numpy(np.stack(poses)))


def make_sequence_cached(batch: int = 1, seq_len: int = 2, height: int = 480, width: int = 640, seed: int = 0, **kw):
    """make_sequence with the result kept under $GS_SYNTH_CACHE (default /tmp/gs_synth_cache): the fp64 ray cast of a
    200-frame 640x480 sequence takes the better part of a minute on the host, and one GPU lease runs the same sequence
    several times (bench, profiles, tests).  The cache holds this function's own output only."""
    import hashlib
    import os

    root = os.environ.get("GS_SYNTH_CACHE", "/tmp/gs_synth_cache")
    key = repr((batch, seq_len, height, width, seed, sorted(kw.items())))
    path = os.path.join(root, "seq_" + hashlib.sha1(key.encode()).hexdigest()[:16] + ".pt")
    if os.path.exists(path):
        try:
            out = torch.load(path, weights_only=True)
            if isinstance(out, (list, tuple)) and len(out) == 4:
                return tuple(out)
        except Exception:
            pass
    out = make_sequence(batch, seq_len, height, width, seed, **kw)
    try:
        os.makedirs(root, exist_ok=True)
        tmp = path + ".%d.tmp" % os.getpid()
        torch.save(list(out), tmp)
        os.replace(tmp, path)
    except OSError:
        pass
    return out
