"""Tensor-level front of the C ABI: one Python function per kernel family, plus the
`torch.autograd.Function`s that keep the reference's gradients flowing through them.

Nothing here computes on the CPU.  Tensors must live on a HIP device (`require_hip`).
Shapes follow the reference: images channels-last (B,L,H,W,C), clouds padded (B,N,C), tables
(P,4) int64 rows [b,n,h,w].
"""
from typing import Optional, Tuple

import torch

from . import _native as nv
from ._native import call, ptr, require_hip, stream, workspace, ws_bytes


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def dev_int(value: int, device) -> torch.Tensor:
    """A device-resident int32 scalar (filled by a kernel: no host->device copy, no sync)."""
    return torch.full((1,), int(value), dtype=torch.int32, device=device)


# ---------------------------------------------------------------------------------------------- V
def vertex_normal_maps_raw(depth, K, poses, want_local=True, want_global=True):
    """depth (B,L,H,W,1), K (B,1,4,4), poses (B,L,4,4) or None -> (V, N, gV, gN); entries not
    requested are None.  One fused launch (reference structures/rgbdimages.py:643-762)."""
    require_hip(depth, K, poses, op="vertex_normal_maps")
    depth, K, poses = _f32c(depth), _f32c(K), _f32c(poses)
    B, L, H, W = depth.shape[:4]
    mk = lambda: torch.empty((B, L, H, W, 3), dtype=torch.float32, device=depth.device)
    V = mk() if want_local else None
    N = mk() if want_local else None
    gV = mk() if want_global else None
    gN = mk() if want_global else None
    call("gs_vertex_normal_maps", ptr(depth), ptr(K), ptr(poses), B, L, H, W, ptr(V), ptr(N), ptr(gV), ptr(gN), stream())
    return V, N, gV, gN


class _MapsFn(torch.autograd.Function):
    """(depth, K, poses) -> (V, N, gV, gN) with the hand-written adjoint."""

    @staticmethod
    def forward(ctx, depth, K, poses, want_local, want_global):
        V, N, gV, gN = vertex_normal_maps_raw(depth, K, poses, want_local, want_global)
        ctx.save_for_backward(depth, K, poses)
        ctx.has_poses = poses is not None
        dummy = depth.new_empty(0)
        outs = tuple(x if x is not None else dummy for x in (V, N, gV, gN))
        ctx.mark_non_differentiable(*[o for o, x in zip(outs, (V, N, gV, gN)) if x is None])
        return outs

    @staticmethod
    def backward(ctx, gV_l, gN_l, gV_g, gN_g):
        depth, K, poses = ctx.saved_tensors
        fix = lambda g: None if (g is None or g.numel() == 0) else g
        g_depth, g_K, g_P = vertex_normal_maps_backward_raw(depth, K, poses, fix(gV_l), fix(gN_l), fix(gV_g), fix(gN_g))
        return g_depth.view_as(depth), g_K.view_as(K), (g_P.view_as(poses) if g_P is not None else None), None, None


def vertex_normal_maps_backward_raw(depth, K, poses, gV_l=None, gN_l=None, gV_g=None, gN_g=None):
    """Adjoint of vertex_normal_maps_raw: given the adjoints of any of the four maps (None = zero) returns
    (g_depth, g_K, g_poses or None), freshly allocated and fully written."""
    depth_c, K_c, poses_c = _f32c(depth), _f32c(K), _f32c(poses)
    B, L, H, W = depth_c.shape[:4]
    gV_l, gN_l, gV_g, gN_g = _f32c(gV_l), _f32c(gN_l), _f32c(gV_g), _f32c(gN_g)
    g_depth = torch.zeros_like(depth_c)
    g_K = torch.zeros_like(K_c)
    g_P = torch.zeros_like(poses_c) if poses_c is not None else None
    ws = workspace(ws_bytes("gs_vertex_normal_maps_backward_ws_bytes", B, L, H, W), depth_c.device, "maps_bwd")
    call("gs_vertex_normal_maps_backward", ptr(depth_c), ptr(K_c), ptr(poses_c), B, L, H, W, ptr(gV_l), ptr(gN_l),
         ptr(gV_g), ptr(gN_g), ptr(g_depth), ptr(g_K), ptr(g_P), ptr(ws), ws.numel(), stream())
    return g_depth, g_K, g_P


def vertex_normal_maps_backward_into(depth, K, poses, gV_l, gN_l, gV_g, gN_g, g_depth, g_K, g_poses):
    """vertex_normal_maps_backward_raw that ADDS into caller-held adjoints (contiguous float32, same shapes as depth / K /
    poses; the C kernels accumulate into all three): what a reverse pass over many frames wants -- no temporaries, no
    zero fills, no `+=` launches per frame."""
    depth_c, K_c, poses_c = _f32c(depth), _f32c(K), _f32c(poses)
    B, L, H, W = depth_c.shape[:4]
    for x in (g_depth, g_K, g_poses):
        if x is not None and not (x.is_contiguous() and x.dtype == torch.float32):
            raise ValueError("vertex_normal_maps_backward_into: the adjoint buffers must be contiguous float32")
    gV_l, gN_l, gV_g, gN_g = _f32c(gV_l), _f32c(gN_l), _f32c(gV_g), _f32c(gN_g)
    ws = workspace(ws_bytes("gs_vertex_normal_maps_backward_ws_bytes", B, L, H, W), depth_c.device, "maps_bwd")
    call("gs_vertex_normal_maps_backward", ptr(depth_c), ptr(K_c), ptr(poses_c), B, L, H, W, ptr(gV_l), ptr(gN_l),
         ptr(gV_g), ptr(gN_g), ptr(g_depth), ptr(g_K), ptr(g_poses), ptr(ws), ws.numel(), stream())


def vertex_normal_maps(depth, K, poses, want_local=True, want_global=True):
    """Autograd-aware entry: returns (V, N, gV, gN) (None where not requested)."""
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (depth, K, poses))
    if not needs_grad:
        return vertex_normal_maps_raw(depth, K, poses, want_local, want_global)
    outs = _MapsFn.apply(depth, K, poses, want_local, want_global)
    sel = (want_local, want_local, want_global, want_global)
    return tuple(o if s else None for o, s in zip(outs, sel))


# ---------------------------------------------------------------------------------------------- alpha
class _AlphaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, sigma, eps):
        p = _f32c(pts)
        out = torch.empty(p.shape[:-1], dtype=torch.float32, device=p.device)
        call("gs_get_alpha", ptr(p), p.numel() // 3, float(sigma), float(eps), ptr(out), stream())
        ctx.save_for_backward(p)
        ctx.sigma, ctx.eps = float(sigma), float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        gp = torch.zeros_like(p)
        call("gs_get_alpha_backward", ptr(p), p.numel() // 3, ctx.sigma, ctx.eps, ptr(_f32c(g)), ptr(gp), stream())
        return gp, None, None


def get_alpha_lastdim(points: torch.Tensor, sigma: float, eps: float) -> torch.Tensor:
    """points (..., 3) -> alpha (...)  (reference slam/fusionutils.py:69-73)."""
    require_hip(points, op="get_alpha")
    return _AlphaFn.apply(points, sigma, eps)


# ---------------------------------------------------------------------------------------------- compaction
def compact_rows(src: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """src (n, C) float32 or (n, C) int64, mask (n,) bool/uint8 -> src[mask], order preserved.
    One host sync to learn the output length."""
    require_hip(src, mask, op="compact_rows")
    n = src.shape[0]
    src = src.contiguous()
    m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
    words = (src.element_size() * (src.numel() // max(n, 1))) // 4 if n else 0
    out = torch.empty_like(src)
    cnt = torch.zeros(1, dtype=torch.int32, device=src.device)
    ws = workspace(ws_bytes("gs_compact_ws_bytes", n), src.device, "compact")
    call("gs_compact_rows", ptr(src), ptr(m8), n, words, ptr(out), ptr(cnt), ptr(ws), ws.numel(), stream())
    return out[: int(cnt.item())]


def compact_multi_raw(srcs, mask):
    """srcs: list of <=4 (n, C_a) float32 tensors sharing `mask` (n,) uint8/bool -> (list of (n, C_a)
    output buffers, count (1,) int32 device).  No sync; rows beyond count are garbage."""
    import ctypes

    require_hip(mask, *srcs, op="compact_multi")
    n = mask.shape[0]
    srcs = [_f32c(s) for s in srcs]
    m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
    outs = [torch.empty_like(s) for s in srcs]
    k = len(srcs)
    a_src = (ctypes.c_void_p * k)(*[s.data_ptr() for s in srcs])
    a_out = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs])
    a_w = (ctypes.c_int * k)(*[s.numel() // max(n, 1) for s in srcs])
    cnt = torch.zeros(1, dtype=torch.int32, device=mask.device)
    ws = workspace(ws_bytes("gs_compact_ws_bytes", n), mask.device, "compact")
    call("gs_compact_multi", k, a_src, a_w, a_out, ptr(m8), n, ptr(cnt), ptr(ws), ws.numel(), stream())
    return outs, cnt


class _MaskSelectFn(torch.autograd.Function):
    """x[mask] on (n, C) rows with a scatter adjoint (the reference's boolean-mask indexing)."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        ctx.shape = x.shape
        return compact_rows(x, mask)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        out = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        out[mask.bool() if mask.dtype != torch.bool else mask] = g  # adjoint of a gather with unique targets
        return out, None


class _MultiMaskSelectFn(torch.autograd.Function):
    """[x[mask] for x in xs] for up to 4 (n, C_a) arrays sharing one mask: one compaction (one scan, one host
    sync for the length) forward, one expansion kernel backward."""

    @staticmethod
    def forward(ctx, mask, *xs):
        outs, cnt = compact_multi_raw(list(xs), mask)
        n = int(cnt.item())
        ctx.save_for_backward(mask)
        ctx.shapes = [x.shape for x in xs]
        return tuple(o[:n] for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        import ctypes

        (mask,) = ctx.saved_tensors
        n = mask.shape[0]
        gs = [_f32c(g) for g in gs]
        dev = mask.device
        outs = [torch.empty(shape, dtype=torch.float32, device=dev) for shape in ctx.shapes]
        if n == 0:
            return (None, *outs)
        m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
        k = len(gs)
        # an empty selection leaves nothing to read: give the kernel a valid (unused) pointer
        a_g = (ctypes.c_void_p * k)(*[(g if g.numel() else o).data_ptr() for g, o in zip(gs, outs)])
        a_out = (ctypes.c_void_p * k)(*[o.data_ptr() for o in outs])
        a_w = (ctypes.c_int * k)(*[o.numel() // n for o in outs])
        ws = workspace(ws_bytes("gs_compact_ws_bytes", n), dev, "compact")
        call("gs_expand_multi", k, a_g, a_w, a_out, ptr(m8), n, ptr(ws), ws.numel(), stream())
        return (None, *outs)


def mask_select_multi(xs, mask: torch.Tensor):
    """[x[mask] for x in xs] (<= 4 arrays, (n, C_a) float32 each), differentiable, one scan and one sync."""
    return list(_MultiMaskSelectFn.apply(mask, *xs))


def select_rows_multi(xs, mask: torch.Tensor):
    """[x[mask] for x in xs] with or without autograd: one compaction, one host sync either way."""
    if torch.is_grad_enabled() and any(x.requires_grad for x in xs):
        return mask_select_multi(xs, mask)
    outs, cnt = compact_multi_raw(list(xs), mask)
    n = int(cnt.item())
    return [o[:n] for o in outs]


def mask_select(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    if torch.is_grad_enabled() and x.requires_grad:
        return _MaskSelectFn.apply(x, mask)
    return compact_rows(x, mask)


def frames_from_raw(depth_u16, rgb_u8, height: int, width: int, depth_scale: float, normalize_color: bool):
    """Raw sensor frames on the device -> float32 (depth (B,H,W,1), rgb (B,H,W,3)); either input may be None.
    depth_u16 (B,Hs,Ws) int16/uint16 storage, rgb_u8 (B,Hs,Ws,3) uint8 (reference datasets/tum.py:455-499)."""
    ref = depth_u16 if depth_u16 is not None else rgb_u8
    require_hip(*(x for x in (depth_u16, rgb_u8) if x is not None), op="frames_from_raw")
    B, Hs, Ws = ref.shape[:3]
    dev = ref.device
    if depth_u16 is not None and not (depth_u16.is_contiguous() and depth_u16.element_size() == 2):
        raise ValueError("frames_from_raw: depth must be a contiguous 16-bit integer tensor")
    if rgb_u8 is not None and not (rgb_u8.is_contiguous() and rgb_u8.dtype == torch.uint8 and rgb_u8.shape[-1] == 3):
        raise ValueError("frames_from_raw: rgb must be a contiguous uint8 (B,H,W,3) tensor")
    depth = torch.empty((B, height, width, 1), dtype=torch.float32, device=dev) if depth_u16 is not None else None
    rgb = torch.empty((B, height, width, 3), dtype=torch.float32, device=dev) if rgb_u8 is not None else None
    call("gs_frames_from_raw", ptr(depth_u16), ptr(rgb_u8), B, Hs, Ws, int(height), int(width), float(depth_scale),
         1 if normalize_color else 0, ptr(depth), ptr(rgb), stream())
    return depth, rgb


# ---------------------------------------------------------------------------------------------- D
def downsample_frame_raw(depth, gV, gN, rgb, ds: int):
    """One-frame maps (B,1,H,W,C) -> padded (B,cap,3) x3 + counts (B,) int32 device
    (reference odometry/icputils.py:651-669)."""
    require_hip(depth, gV, gN, rgb, op="downsample_frame")
    depth, gV, gN, rgb = _f32c(depth), _f32c(gV), _f32c(gN), _f32c(rgb)
    B, _, H, W = depth.shape[:4]
    cap = ((H + ds - 1) // ds) * ((W + ds - 1) // ds)
    dev = depth.device
    mk = lambda src: None if src is None else torch.zeros((B, cap, 3), dtype=torch.float32, device=dev)
    op, on, oc = mk(gV), mk(gN), mk(rgb)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    ws = workspace(ws_bytes("gs_downsample_frame_ws_bytes", H, W, ds), dev, "compact")
    call("gs_downsample_frame", ptr(depth), ptr(gV), ptr(gN), ptr(rgb), B, H, W, ds, cap, ptr(op), ptr(on), ptr(oc),
         None, ptr(counts), ptr(ws), ws.numel(), stream())
    return op, on, oc, counts


# ---------------------------------------------------------------------------------------------- P / S
def project_active_raw(points_padded, counts_dev, poses_b44, K_b44, H: int, W: int, ds: int = 0):
    """-> (rows buffer (B*Nmax,4) int64, count (1,) int32 device).  Rows beyond count are garbage.
    reference slam/fusionutils.py:247-282 (+ odometry/icputils.py:596-597 when ds > 0)."""
    require_hip(points_padded, counts_dev, poses_b44, K_b44, op="project_active")
    pts, poses, K = _f32c(points_padded), _f32c(poses_b44), _f32c(K_b44)
    B, Nmax = pts.shape[:2]
    dev = pts.device
    rows = torch.empty((B * Nmax, 4), dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = workspace(ws_bytes("gs_project_active_ws_bytes", B, Nmax), dev, "project")
    call("gs_project_active", ptr(pts), ptr(counts_dev), B, Nmax, ptr(poses), ptr(K), H, W, ds, ptr(rows), ptr(cnt),
         ptr(ws), ws.numel(), stream())
    return rows, cnt


def gather_table_rows_raw(rows, n_rows_dev, max_rows: int, attr_padded, cap: int):
    """Per-batch gather of attr[b, n] for table rows (sorted by b) -> padded (B,cap,C), counts (B,)."""
    require_hip(rows, attr_padded, op="gather_table_rows")
    attr = _f32c(attr_padded)
    B, Nmax, C = attr.shape
    dev = attr.device
    out = torch.zeros((B, max(cap, 1), C), dtype=torch.float32, device=dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    ws = workspace(ws_bytes("gs_gather_table_rows_ws_bytes", B), dev, "gather")
    call("gs_gather_table_rows", ptr(rows), ptr(n_rows_dev), max_rows, ptr(attr), B, Nmax, C, cap, ptr(out), ptr(counts),
         ptr(ws), ws.numel(), stream())
    return out, counts


def table_ds_mask(rows: torch.Tensor, ds: int) -> torch.Tensor:
    require_hip(rows, op="table_ds_mask")
    rows = rows.contiguous()
    mask = torch.zeros(rows.shape[0], dtype=torch.uint8, device=rows.device)
    call("gs_table_ds_mask", ptr(rows), rows.shape[0], ds, ptr(mask), stream())
    return mask


# ---------------------------------------------------------------------------------------------- K / J
def knn1_raw(src, tgt, ns_dev=None, nt_dev=None, brute_force: bool = False) -> torch.Tensor:
    """src (Ns,3), tgt (Nt,3) -> packed best (Ns,) int64 = dist_bits << 32 | idx.  The default kernel
    prunes target chunks with an exact AABB bound; brute_force=True evaluates every pair (verifier)."""
    require_hip(src, tgt, op="knn1")
    src, tgt = _f32c(src), _f32c(tgt)
    dev = src.device
    ns_dev = dev_int(src.shape[0], dev) if ns_dev is None else ns_dev
    nt_dev = dev_int(tgt.shape[0], dev) if nt_dev is None else nt_dev
    best = torch.empty(src.shape[0], dtype=torch.int64, device=dev)
    if brute_force:
        call("gs_knn1_bruteforce", ptr(src), ptr(ns_dev), src.shape[0], ptr(tgt), ptr(nt_dev), tgt.shape[0], ptr(best), stream())
    else:
        ws = workspace(ws_bytes("gs_knn1_ws_bytes", tgt.shape[0]), dev, "knn")
        call("gs_knn1", ptr(src), ptr(ns_dev), src.shape[0], ptr(tgt), ptr(nt_dev), tgt.shape[0], ptr(best), ptr(ws),
             ws.numel(), stream())
    return best


def knn1_unpack(best: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    dev = best.device
    n = best.shape[0]
    d2 = torch.empty(n, dtype=torch.float32, device=dev)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    call("gs_knn1_unpack", ptr(best), ptr(dev_int(n, dev)), n, ptr(d2), ptr(idx), stream())
    return d2, idx


def knn_points(src: torch.Tensor, tgt: torch.Tensor):
    """Drop-in for the call `knn_points(src (1,Ns,3), tgt (1,Nt,3))` of reference
    odometry/icputils.py:200-201: returns (dists (1,Ns,1) squared L2, idx (1,Ns,1) int64)."""
    d2, idx = knn1_unpack(knn1_raw(src[0].detach(), tgt[0].detach()))
    return d2.view(1, -1, 1), idx.view(1, -1, 1)


def _thresh(dist_thresh) -> float:
    # negative == None at the C ABI; a user-supplied negative threshold keeps nothing, like 0.0
    return -1.0 if dist_thresh is None else max(float(dist_thresh), 0.0)


def icp_linearize_raw(src, tgt, nrm, best, dist_thresh) -> torch.Tensor:
    """-> 44 floats: H (36) | g (6) | e | count."""
    src, tgt, nrm = _f32c(src), _f32c(tgt), _f32c(nrm)
    dev = src.device
    out = torch.empty(44, dtype=torch.float32, device=dev)
    ws = workspace(ws_bytes("gs_icp_linearize_ws_bytes", src.shape[0]), dev, "linearize")
    call("gs_icp_linearize", ptr(src), ptr(dev_int(src.shape[0], dev)), src.shape[0], ptr(tgt), ptr(nrm), ptr(best),
         _thresh(dist_thresh), ptr(out), ptr(ws), ws.numel(), stream())
    return out


class _LinearizeFn(torch.autograd.Function):
    """(src, tgt, nrm | best) -> (H (6,6), g (6,1), e ()) with the scatter adjoint of A.5."""

    @staticmethod
    def forward(ctx, src, tgt, nrm, best, dist_thresh):
        out = icp_linearize_raw(src, tgt, nrm, best, dist_thresh)
        ctx.save_for_backward(src, tgt, nrm, best)
        ctx.dist_thresh = dist_thresh
        return out[:36].view(6, 6).clone(), out[36:42].view(6, 1).clone(), out[42].clone()

    @staticmethod
    def backward(ctx, gH, gg, ge):
        src, tgt, nrm, best = ctx.saved_tensors
        src_c, tgt_c, nrm_c = _f32c(src), _f32c(tgt), _f32c(nrm)
        dev = src.device
        z = lambda t, n: torch.zeros(n, dtype=torch.float32, device=dev) if t is None else _f32c(t).reshape(-1)
        gout = torch.cat([z(gH, 36), z(gg, 6), z(ge, 1)])
        g_src = torch.zeros_like(src_c)
        g_tgt = torch.zeros_like(tgt_c)
        g_nrm = torch.zeros_like(nrm_c)
        call("gs_icp_linearize_backward", ptr(src_c), ptr(dev_int(src_c.shape[0], dev)), src_c.shape[0], ptr(tgt_c),
             ptr(nrm_c), ptr(best), _thresh(ctx.dist_thresh), ptr(gout), ptr(g_src), ptr(g_tgt), ptr(g_nrm), stream())
        return g_src, g_tgt, g_nrm, None, None


def icp_linearize(src, tgt, nrm, best, dist_thresh):
    return _LinearizeFn.apply(src, tgt, nrm, best, dist_thresh)


def icp_rows_raw(src, tgt, nrm, best, dist_thresh):
    """-> A (Ns,6), b (Ns,), keep (Ns,) uint8 (rows failing the distance filter are zero)."""
    src, tgt, nrm = _f32c(src), _f32c(tgt), _f32c(nrm)
    dev = src.device
    n = src.shape[0]
    A = torch.empty((n, 6), dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    call("gs_icp_rows", ptr(src), ptr(dev_int(n, dev)), n, ptr(tgt), ptr(nrm), ptr(best), _thresh(dist_thresh), ptr(A),
         ptr(b), ptr(keep), stream())
    return A, b, keep


class _TransformFn(torch.autograd.Function):
    """(pts (N,3), T (4,4)) -> R pts + t (reference geometry/geometryutils.py:780-792)."""

    @staticmethod
    def forward(ctx, pts, T):
        p, Tc = _f32c(pts), _f32c(T)
        out = torch.empty_like(p)
        call("gs_transform_points", ptr(p), ptr(dev_int(p.shape[0], p.device)), p.shape[0], ptr(Tc), ptr(out), stream())
        ctx.save_for_backward(p, Tc)
        return out

    @staticmethod
    def backward(ctx, g):
        p, T = ctx.saved_tensors
        g = _f32c(g)
        # adjoint w.r.t. the points is the transform by R^T (no translation): same kernel, T' = [R^T 0]
        Tt = torch.zeros_like(T)
        Tt[:3, :3] = T[:3, :3].t()
        Tt[3, 3] = 1.0
        gp = torch.empty_like(p)
        call("gs_transform_points", ptr(g), ptr(dev_int(p.shape[0], p.device)), p.shape[0], ptr(Tt.contiguous()), ptr(gp), stream())
        gT = torch.zeros_like(T)
        gT[:3, :3] = g.t() @ p  # 3xN @ Nx3: a library GEMM, O(N) once per call
        gT[:3, 3] = g.sum(0)
        return gp, gT


def transform_points(pts: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    require_hip(pts, T, op="transform_points")
    return _TransformFn.apply(pts, T)


# ---------------------------------------------------------------------------------------------- X
def icp_device_loop(src, tgt, nrm, init_T, numiters, damp, dist_thresh, grad_params=None, want_trace=False,
                    want_best=False):
    """Whole (grad)ICP loop on the device, no host sync (reference odometry/icputils.py:310-367,
    :479-545).  Returns (T (4,4), best_last or None, trace (numiters,48) or None)."""
    require_hip(src, tgt, nrm, init_T, op="icp")
    src, tgt, nrm, init_T = _f32c(src.detach()), _f32c(tgt.detach()), _f32c(nrm.detach()), _f32c(init_T.detach())
    dev = src.device
    ns, nt = src.shape[0], tgt.shape[0]
    if ns == 0 or nt == 0:
        raise ValueError("ICP needs non-empty source and target clouds (got {} and {} points)".format(ns, nt))
    T = torch.empty((4, 4), dtype=torch.float32, device=dev)
    best = torch.empty(ns, dtype=torch.int64, device=dev) if want_best else None
    trace = torch.zeros((max(numiters, 1), 48), dtype=torch.float32, device=dev) if want_trace else None
    ws = workspace(ws_bytes("gs_icp_ws_bytes", ns, nt), dev, "icp")
    d_ns, d_nt = dev_int(ns, dev), dev_int(nt, dev)
    if grad_params is None:
        call("gs_icp_point_to_plane", ptr(src), ptr(d_ns), ns, ptr(tgt), ptr(nrm), ptr(d_nt), nt, ptr(init_T),
             int(numiters), float(damp), _thresh(dist_thresh), None, ptr(T), ptr(best), ptr(trace), ptr(ws), ws.numel(), stream())
    else:
        lmax, Bp, B2, nu = grad_params
        call("gs_icp_point_to_plane_grad", ptr(src), ptr(d_ns), ns, ptr(tgt), ptr(nrm), ptr(d_nt), nt, ptr(init_T),
             int(numiters), float(damp), _thresh(dist_thresh), float(lmax), float(Bp), float(B2), float(nu), None, ptr(T),
             ptr(best), ptr(trace), ptr(ws), ws.numel(), stream())
    return T, best, trace


class _IcpLoopFn(torch.autograd.Function):
    """Differentiable point_to_plane_ICP / gradICP as ONE node: the taped device loop forward, the
    device-side reverse pass over the tape backward (no host sync in either direction)."""

    @staticmethod
    def forward(ctx, src, tgt, nrm, init_T, numiters, damp, dist_thresh, grad_params):
        src, tgt, nrm, init_T = _f32c(src.detach()), _f32c(tgt.detach()), _f32c(nrm.detach()), _f32c(init_T.detach())
        dev = src.device
        ns, nt = src.shape[0], tgt.shape[0]
        if ns == 0 or nt == 0:
            raise ValueError("ICP needs non-empty source and target clouds (got {} and {} points)".format(ns, nt))
        grad_lm = 1 if grad_params is not None else 0
        lmax, Bp, B2, nu = grad_params if grad_params is not None else (2.0, 1.0, 1.0, 200.0)
        T = torch.empty((4, 4), dtype=torch.float32, device=dev)
        best = torch.empty(ns, dtype=torch.int64, device=dev)
        tape = torch.empty(ws_bytes("gs_icp_tape_bytes", ns, int(numiters), grad_lm), dtype=torch.uint8, device=dev)
        ws = workspace(ws_bytes("gs_icp_ws_bytes", ns, nt), dev, "icp")
        d_ns, d_nt = dev_int(ns, dev), dev_int(nt, dev)
        call("gs_icp_point_to_plane_taped", ptr(src), ptr(d_ns), ns, ptr(tgt), ptr(nrm), ptr(d_nt), nt, ptr(init_T),
             int(numiters), float(damp), _thresh(dist_thresh), grad_lm, float(lmax), float(Bp), float(B2), float(nu), None,
             ptr(T), ptr(best), ptr(tape), tape.numel(), ptr(ws), ws.numel(), stream())
        ctx.save_for_backward(src, tgt, nrm, init_T, tape, d_ns, d_nt)
        ctx.cfg = (int(numiters), _thresh(dist_thresh), grad_lm, float(lmax), float(Bp), float(B2), float(nu))
        ctx.mark_non_differentiable(best)
        return T, best

    @staticmethod
    def backward(ctx, gT, _gbest):
        src, tgt, nrm, init_T, tape, d_ns, d_nt = ctx.saved_tensors
        numiters, thresh, grad_lm, lmax, Bp, B2, nu = ctx.cfg
        dev = src.device
        ns, nt = src.shape[0], tgt.shape[0]
        gT = _f32c(gT)
        need_tgt, need_nrm = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        g_src = torch.empty_like(src)
        g_tgt = torch.empty_like(tgt) if need_tgt else None
        g_nrm = torch.empty_like(nrm) if need_nrm else None
        g_init = torch.empty((4, 4), dtype=torch.float32, device=dev)
        ws = workspace(ws_bytes("gs_icp_backward_ws_bytes", ns), dev, "icp_bwd")
        call("gs_icp_point_to_plane_backward", ptr(src), ptr(d_ns), ns, ptr(tgt), ptr(nrm), ptr(d_nt), nt, ptr(init_T), numiters, thresh,
             grad_lm, lmax, Bp, B2, nu, ptr(tape), tape.numel(), ptr(gT), ptr(g_src), ptr(g_tgt), ptr(g_nrm), ptr(g_init),
             ptr(ws), ws.numel(), stream())
        return g_src, g_tgt, g_nrm, g_init, None, None, None, None


def icp_loop_autograd(src, tgt, nrm, init_T, numiters, damp, dist_thresh, grad_params=None):
    """(T (4,4) with grad_fn, packed NN of the last iteration's first solve)."""
    require_hip(src, tgt, nrm, init_T, op="icp")
    return _IcpLoopFn.apply(src, tgt, nrm, init_T, numiters, damp, dist_thresh, grad_params)


def slam_localize_raw(depth, K, prev_poses, map_points, map_normals, map_counts_i32, ds, numiters, damp, dist_thresh,
                      grad_params=None, out=None, want_maps=True):
    """One fused, sync-free ICPSLAM._localize (reference slam/icpslam.py:238-247).
    depth (B,1,H,W,1), K (B,1,4,4), prev_poses (B,1,4,4), map padded (B,Nmax,3) x2 + counts (B,) int32.
    Returns (poses (B,1,4,4), V, N): the local maps are handed back so the caller can cache them (want_maps=False: not
    computed, None returned).  `out`: a contiguous float32 (B,1,4,4) tensor the poses are written to (a sequence driver's
    slice of its pose array: no copy launch)."""
    require_hip(depth, K, prev_poses, map_points, map_normals, map_counts_i32, op="slam_localize")
    depth, K, prev = _f32c(depth.detach()), _f32c(K.detach()), _f32c(prev_poses.detach())
    mp, mn = _f32c(map_points.detach()), _f32c(map_normals.detach())
    B, _, H, W = depth.shape[:4]
    Nmax = mp.shape[1]
    dev = depth.device
    mk = lambda: torch.empty((B, 1, H, W, 3), dtype=torch.float32, device=dev)
    V, N, gV = (mk() if want_maps else None), (mk() if want_maps else None), mk()
    if out is None:
        out = torch.empty((B, 1, 4, 4), dtype=torch.float32, device=dev)
    elif not (out.is_contiguous() and out.dtype == torch.float32 and out.numel() == 16 * B and out.device == dev):
        raise ValueError("slam_localize: `out` must be a contiguous float32 (B,1,4,4) tensor on the inputs' device")
    ws = workspace(ws_bytes("gs_slam_localize_ws_bytes", B, H, W, int(ds), Nmax), dev, "localize")
    lmax, Bp, B2, nu = grad_params if grad_params is not None else (2.0, 1.0, 1.0, 200.0)
    call("gs_slam_localize", ptr(depth), ptr(K), ptr(prev), B, H, W, int(ds), ptr(mp), ptr(mn), ptr(map_counts_i32), Nmax,
         1 if grad_params is not None else 0, int(numiters), float(damp), _thresh(dist_thresh), float(lmax), float(Bp),
         float(B2), float(nu), ptr(V), ptr(N), ptr(gV), None, ptr(out), ptr(ws), ws.numel(), stream())
    return out, V, N


class _LocalizeFn(torch.autograd.Function):
    """Differentiable ICPSLAM._localize as ONE node (gs_slam_localize_taped / _backward): gradients reach the
    live frame's global vertex map, the map points / normals that served as ICP targets and the previous
    pose; nothing synchronises with the host in either direction."""

    @staticmethod
    def forward(ctx, gV, depth, K, prev_poses, mp, mn, counts_i32, ds, numiters, damp, dist_thresh, grad_params):
        gV, depth, K, prev = _f32c(gV.detach()), _f32c(depth.detach()), _f32c(K.detach()), _f32c(prev_poses.detach())
        mp, mn = _f32c(mp.detach()), _f32c(mn.detach())
        B, _, H, W = depth.shape[:4]
        Nmax = mp.shape[1]
        dev = depth.device
        grad_lm = 1 if grad_params is not None else 0
        lmax, Bp, B2, nu = grad_params if grad_params is not None else (2.0, 1.0, 1.0, 200.0)
        out = torch.empty((B, 1, 4, 4), dtype=torch.float32, device=dev)
        tape = torch.empty(ws_bytes("gs_slam_localize_tape_bytes", B, H, W, int(ds), Nmax, int(numiters), grad_lm),
                           dtype=torch.uint8, device=dev)
        ws = workspace(ws_bytes("gs_slam_localize_ws_bytes", B, H, W, int(ds), Nmax), dev, "localize")
        call("gs_slam_localize_taped", ptr(depth), ptr(gV), ptr(K), ptr(prev), B, H, W, int(ds), ptr(mp), ptr(mn),
             ptr(counts_i32), Nmax, grad_lm, int(numiters), float(damp), _thresh(dist_thresh), float(lmax), float(Bp), float(B2),
             float(nu), ptr(out), ptr(tape), tape.numel(), ptr(ws), ws.numel(), stream())
        ctx.save_for_backward(prev, mp, mn, tape)
        ctx.cfg = (B, H, W, int(ds), Nmax, grad_lm, int(numiters), _thresh(dist_thresh), float(lmax), float(Bp), float(B2), float(nu))
        return out

    @staticmethod
    def backward(ctx, g_out):
        prev, mp, mn, tape = ctx.saved_tensors
        B, H, W, ds, Nmax, grad_lm, numiters, thresh, lmax, Bp, B2, nu = ctx.cfg
        dev = prev.device
        g_out = _f32c(g_out)
        g_gV = torch.empty((B, 1, H, W, 3), dtype=torch.float32, device=dev)
        g_mp = torch.empty_like(mp) if ctx.needs_input_grad[4] else None
        g_mn = torch.empty_like(mn) if ctx.needs_input_grad[5] else None
        g_prev = torch.empty((B, 1, 4, 4), dtype=torch.float32, device=dev)
        ws = workspace(ws_bytes("gs_slam_localize_backward_ws_bytes", B, H, W, ds, Nmax), dev, "localize_bwd")
        call("gs_slam_localize_backward", ptr(prev), B, H, W, ds, ptr(mp), ptr(mn), Nmax, grad_lm, numiters, thresh, lmax, Bp, B2,
             nu, ptr(tape), tape.numel(), ptr(g_out), ptr(g_gV), ptr(g_mp), ptr(g_mn), ptr(g_prev), 0, ptr(ws), ws.numel(), stream())
        return g_gV, None, None, g_prev, g_mp, g_mn, None, None, None, None, None, None


def slam_localize_autograd(gV, depth, K, prev_poses, map_points, map_normals, map_counts_i32, ds, numiters, damp, dist_thresh,
                           grad_params=None):
    """Differentiable fused localisation -> poses (B,1,4,4) with grad_fn."""
    require_hip(gV, depth, K, prev_poses, map_points, map_normals, map_counts_i32, op="slam_localize")
    return _LocalizeFn.apply(gV, depth, K, prev_poses, map_points, map_normals, map_counts_i32, ds, numiters, damp, dist_thresh,
                             grad_params)


def pointfusion_update_raw(depth, rgb, K, poses, map_points, map_normals, map_colors, map_ccounts, map_counts_i32, dist_th,
                           dot_th, sigma, stats=None):
    """One fused, sync-free PointFusion map update on arena arrays (reference slam/fusionutils.py:761-789 with
    inplace=True).  depth (B,H,W[,1]), rgb (B,H,W,3), K / poses (B,[1,]4,4); map_* (B,Nmax,C) float32 contiguous,
    rows beyond map_counts zero; the caller guarantees counts + H*W <= Nmax.  Everything is updated in place,
    including map_counts_i32 (B,) int32 on the device; `stats` (4+B,) int32 receives the step's counters."""
    require_hip(depth, rgb, K, poses, map_points, map_normals, map_colors, map_ccounts, map_counts_i32, op="pointfusion_update")
    for name, x in (("map_points", map_points), ("map_normals", map_normals), ("map_colors", map_colors),
                    ("map_ccounts", map_ccounts), ("depth", depth), ("rgb", rgb)):
        if not (x.is_contiguous() and x.dtype == torch.float32):
            raise ValueError("pointfusion_update: {} must be contiguous float32 (it is updated / read in place)".format(name))
    B, H, W = depth.shape[:3]
    Nmax = map_points.shape[1]
    K, poses = _f32c(K), _f32c(poses)
    ws = workspace(ws_bytes("gs_pointfusion_update_ws_bytes", B, H, W, Nmax), depth.device, "fusion_step")
    call("gs_pointfusion_update", ptr(depth), ptr(rgb), ptr(K), ptr(poses), B, H, W, ptr(map_points), ptr(map_normals),
         ptr(map_colors), ptr(map_ccounts), ptr(map_counts_i32), Nmax, float(dist_th), float(dot_th), float(sigma), ptr(stats),
         ptr(ws), ws.numel(), stream())


def aggregate_update_raw(depth, rgb, K, poses, map_points, map_normals, map_colors, map_counts_i32, stats=None):
    """One fused, sync-free ICPSLAM map update on arena arrays (reference slam/fusionutils.py:725-758 with
    inplace=True): every valid live-frame pixel is appended, counts advance on the device."""
    require_hip(depth, rgb, K, poses, map_points, map_normals, map_colors, map_counts_i32, op="aggregate_update")
    for name, x in (("map_points", map_points), ("map_normals", map_normals), ("map_colors", map_colors), ("depth", depth), ("rgb", rgb)):
        if not (x.is_contiguous() and x.dtype == torch.float32):
            raise ValueError("aggregate_update: {} must be contiguous float32 (it is updated / read in place)".format(name))
    B, H, W = depth.shape[:3]
    K, poses = _f32c(K), _f32c(poses)
    ws = workspace(ws_bytes("gs_aggregate_update_ws_bytes", B, H, W), depth.device, "aggregate_step")
    call("gs_aggregate_update", ptr(depth), ptr(rgb), ptr(K), ptr(poses), B, H, W, ptr(map_points), ptr(map_normals),
         ptr(map_colors), ptr(map_counts_i32), map_points.shape[1], ptr(stats), ptr(ws), ws.numel(), stream())


class _PointFusionSeqFn(torch.autograd.Function):
    """PointFusion over a whole batch of sequences as ONE autograd node.

    forward  = the arena-backed frame loop with the taped forms of its two calls per frame
               (gs_slam_localize_taped, gs_pointfusion_update_taped): no host synchronisation until the map is
               handed back, the map updated in place.
    backward = the frames in reverse on a copy of the final arena: per frame gs_pointfusion_update_backward (pulls the
               running adjoint of the whole map back through merge + append at the matched / appended rows only and
               restores the arena to the previous frame's), the adjoint of the frame's vertex / normal maps, then
               gs_slam_localize_backward (adds the ICP targets' adjoints into the running map adjoint) and the adjoint of
               the live maps under the previous pose.  What is a constant in the reference's graph is a constant here
               (correspondence tables, association indices, accept decisions).
    reference: torch autograd through slam/icpslam.py:125-137 with slam/pointfusion.py:107-112."""

    @staticmethod
    def forward(ctx, rgb, depth, K, poses, cfg):
        odom, ds, numiters, damp, dist_thresh, gparams, dist_th, dot_th, sigma, arena_cls = cfg
        rgb, depth, K = _f32c(rgb.detach()), _f32c(depth.detach()), _f32c(K.detach())
        poses_c = _f32c(poses.detach()) if poses is not None else None
        B, L, H, W = depth.shape[:4]
        dev = depth.device
        arena = arena_cls(B, H * W, dev, with_features=True)
        recovered = torch.empty((B, L, 4, 4), dtype=torch.float32, device=dev)
        stats = torch.zeros((L, 4 + B), dtype=torch.int32, device=dev)
        grad_lm = 1 if gparams is not None else 0
        lmax, Bp, B2, nu = gparams if gparams is not None else (2.0, 1.0, 1.0, 200.0)
        fuse_tape_b = ws_bytes("gs_pointfusion_update_tape_bytes", B, H, W)
        frames = []
        prev, bound = None, None
        for s in range(L):
            d_s, c_s = depth[:, s].contiguous(), rgb[:, s].contiguous()
            rec = {}
            if s == 0 or odom == "gt":
                pose = (poses_c[:, s:s + 1].contiguous() if poses_c is not None else
                        torch.eye(4, dtype=torch.float32, device=dev).view(1, 1, 4, 4).repeat(B, 1, 1, 1))
            else:
                mp, mn, _, _ = arena.rows(bound)
                _, _, gV, _ = vertex_normal_maps_raw(d_s.unsqueeze(1), K, prev, want_local=False, want_global=True)
                pose = torch.empty((B, 1, 4, 4), dtype=torch.float32, device=dev)
                tape = torch.empty(ws_bytes("gs_slam_localize_tape_bytes", B, H, W, int(ds), bound, int(numiters), grad_lm),
                                   dtype=torch.uint8, device=dev)
                ws = workspace(ws_bytes("gs_slam_localize_ws_bytes", B, H, W, int(ds), bound), dev, "localize")
                call("gs_slam_localize_taped", ptr(d_s), ptr(gV), ptr(K), ptr(prev), B, H, W, int(ds), ptr(mp), ptr(mn),
                     ptr(arena.counts), bound, grad_lm, int(numiters), float(damp), _thresh(dist_thresh), float(lmax), float(Bp),
                     float(B2), float(nu), ptr(pose), ptr(tape), tape.numel(), ptr(ws), ws.numel(), stream())
                rec["loc"] = (tape, bound)
            bound = arena.reserve_frame()
            mp, mn, mc, mf = arena.rows(bound)
            ftape = torch.empty(fuse_tape_b, dtype=torch.uint8, device=dev)
            ws = workspace(ws_bytes("gs_pointfusion_update_ws_bytes", B, H, W, bound), dev, "fusion_step")
            call("gs_pointfusion_update_taped", ptr(d_s), ptr(c_s), ptr(K), ptr(pose), B, H, W, ptr(mp), ptr(mn), ptr(mc), ptr(mf),
                 ptr(arena.counts), bound, float(dist_th), float(dot_th), float(sigma), ptr(stats[s]), ptr(ftape), ftape.numel(),
                 ptr(ws), ws.numel(), stream())
            arena.appended()
            rec["fuse"] = (ftape, bound)
            frames.append(rec)
            recovered[:, s] = pose[:, 0]
            prev = pose
        n = arena.counts.tolist()  # the one host synchronisation of the sequence
        ctx.arena, ctx.frames, ctx.n_final = arena, frames, n
        ctx.cfg = (odom, int(ds), int(numiters), _thresh(dist_thresh), grad_lm, float(lmax), float(Bp), float(B2), float(nu), float(sigma))
        ctx.has_poses = poses is not None
        ctx.save_for_backward(rgb, depth, K, recovered)
        ctx.stats = stats
        cut = lambda x: x[:, : max(n)].clone()  # zero-padded to the largest map (rows beyond a sequence's count are zero)
        ctx.mark_non_differentiable(stats)
        return cut(arena.points), cut(arena.normals), cut(arena.colors), cut(arena.ccounts), recovered.clone(), stats

    @staticmethod
    def backward(ctx, g_p, g_n, g_c, g_f, g_poses, _g_stats):
        rgb, depth, K, recovered = ctx.saved_tensors
        odom, ds, numiters, thresh, grad_lm, lmax, Bp, B2, nu, sigma = ctx.cfg
        arena, frames, n_final = ctx.arena, ctx.frames, ctx.n_final
        B, L, H, W = depth.shape[:4]
        dev = depth.device
        cap = arena.cap
        # work on copies: the arena is restored frame by frame, the running adjoint of the map is pulled back in place
        mp, mn, mc, mf = (x.clone() for x in (arena.points, arena.normals, arena.colors, arena.ccounts))
        counts = arena.counts.clone()

        def running(g, c):
            G = torch.zeros((B, cap, c), dtype=torch.float32, device=dev)
            if g is not None:
                for b in range(B):  # (rows beyond a sequence's count are padding of the output, not map points)
                    G[b, :n_final[b]] = g[b, :n_final[b]]
            return G

        Gp, Gn, Gc, Gf = running(g_p, 3), running(g_n, 3), running(g_c, 3), running(g_f, 1)
        gpose = g_poses.clone().float() if g_poses is not None else torch.zeros((B, L, 4, 4), dtype=torch.float32, device=dev)
        g_rgb, g_depth, g_K = torch.zeros_like(rgb), torch.zeros_like(depth), torch.zeros_like(K)
        g_poses_in = torch.zeros((B, L, 4, 4), dtype=torch.float32, device=dev) if ctx.has_poses else None
        mk = lambda c: torch.empty((B, H, W, c), dtype=torch.float32, device=dev)
        g_V, g_gV, g_gN = mk(3), mk(3), mk(3)
        g_live = torch.empty((B, 1, H, W, 3), dtype=torch.float32, device=dev)
        g_prev = torch.empty((B, 1, 4, 4), dtype=torch.float32, device=dev)
        # One sequence (B = 1): the per-frame slices of the adjoints are contiguous, so every kernel writes or ADDS straight
        # into them -- no per-frame temporaries, zero fills or `+=` launches (a dozen small torch kernels per frame, and
        # their host cost, in the first version of this loop).  A batch takes contiguous per-frame buffers and copies.
        one = B == 1
        if not one:
            t_rgb, t_depth, t_pose, t_pose2 = mk(3), torch.empty((B, 1, H, W, 1), dtype=torch.float32, device=dev), \
                torch.empty((B, 1, 4, 4), dtype=torch.float32, device=dev), torch.empty((B, 1, 4, 4), dtype=torch.float32, device=dev)
        stride = cap

        def restride(n):
            # A batch's arrays are (B, stride, C) and the kernels take ONE number for row bound and row stride.  The arena
            # may have grown during the forward pass; the tapes of earlier frames were laid out with the stride of their
            # time.  Walking back, the map only shrinks: repack the working arrays to the older, smaller stride once per
            # growth step (rows beyond it did not exist yet; their adjoints have been consumed by their own frames).
            nonlocal mp, mn, mc, mf, Gp, Gn, Gc, Gf, stride
            if one or n == stride:
                return
            mp, mn, mc, mf, Gp, Gn, Gc, Gf = (x[:, :n].contiguous() for x in (mp, mn, mc, mf, Gp, Gn, Gc, Gf))
            stride = n

        for s in reversed(range(L)):
            d_s, c_s = depth[:, s].contiguous(), rgb[:, s].contiguous()
            pose_s = recovered[:, s:s + 1].contiguous()
            ftape, bound = frames[s]["fuse"]
            restride(bound)
            ws = workspace(ws_bytes("gs_pointfusion_update_backward_ws_bytes", B, H, W), dev, "fusion_bwd")
            call("gs_pointfusion_update_backward", ptr(d_s), ptr(c_s), ptr(K), ptr(pose_s), B, H, W, ptr(mp), ptr(mn), ptr(mc), ptr(mf),
                 ptr(counts), bound, sigma, ptr(ftape), ftape.numel(), ptr(Gp), ptr(Gn), ptr(Gc), ptr(Gf), ptr(g_V), ptr(g_gV),
                 ptr(g_gN), ptr(g_rgb[:, s] if one else t_rgb), ptr(ws), ws.numel(), stream())
            if one:
                gd_s, gpose_s = g_depth[:, s], gpose[:, s]
            else:
                g_rgb[:, s] = t_rgb
                gd_s, gpose_s = t_depth.zero_(), t_pose.zero_()
            vertex_normal_maps_backward_into(d_s.unsqueeze(1), K, pose_s, g_V.unsqueeze(1), None, g_gV.unsqueeze(1), g_gN.unsqueeze(1),
                                             gd_s, g_K, gpose_s)
            if not one:
                gpose[:, s] += t_pose[:, 0]
            if "loc" in frames[s]:
                tape, nmax = frames[s]["loc"]
                restride(nmax)
                prev = recovered[:, s - 1:s].contiguous()
                ws = workspace(ws_bytes("gs_slam_localize_backward_ws_bytes", B, H, W, ds, nmax), dev, "localize_bwd")
                call("gs_slam_localize_backward", ptr(prev), B, H, W, ds, ptr(mp), ptr(mn), nmax, grad_lm, numiters, thresh, lmax, Bp, B2,
                     nu, ptr(tape), tape.numel(), ptr(gpose[:, s:s + 1].contiguous()), ptr(g_live), ptr(Gp), ptr(Gn), ptr(g_prev), 1,
                     ptr(ws), ws.numel(), stream())
                gprev_s = gpose[:, s - 1] if one else t_pose2.zero_()
                vertex_normal_maps_backward_into(d_s.unsqueeze(1), K, prev, None, None, g_live, None, gd_s, g_K, gprev_s)
                if not one:
                    gpose[:, s - 1] += t_pose2[:, 0]
                gpose[:, s - 1] += g_prev.view(B, 4, 4)
            elif g_poses_in is not None:
                g_poses_in[:, s] += gpose[:, s]
            if not one:
                g_depth[:, s] += t_depth[:, 0]
        return g_rgb, g_depth, g_K, g_poses_in, None


def pointfusion_sequence_autograd(rgb, depth, K, poses, cfg):
    """-> (points, normals, colors, ccounts (1, N, C) each, recovered poses (1, L, 4, 4), stats) with grad_fn."""
    require_hip(rgb, depth, K, poses, op="pointfusion_sequence")
    return _PointFusionSeqFn.apply(rgb, depth, K, poses, cfg)


# ---------------------------------------------------------------------------------------------- C / U / F / A
def fusion_similar_raw(rows, n_rows_dev, max_rows, gV, gN, map_points, map_normals, dist_th, dot_th):
    """-> keep (max_rows,) uint8, max_dot (1,) float32 device."""
    gV, gN, mp, mn = _f32c(gV), _f32c(gN), _f32c(map_points), _f32c(map_normals)
    H, W = gV.shape[2:4]
    dev = gV.device
    keep = torch.zeros(max(max_rows, 1), dtype=torch.uint8, device=dev)
    max_dot = torch.zeros(1, dtype=torch.float32, device=dev)
    call("gs_fusion_similar", ptr(rows), ptr(n_rows_dev), max_rows, ptr(gV), ptr(gN), H, W, ptr(mp), ptr(mn),
         mp.shape[1], float(dist_th), float(dot_th), ptr(keep), ptr(max_dot), stream())
    return keep, max_dot


def fusion_unique_raw(rows, keep, n_rows_dev, max_rows, gV, map_points, map_ccounts):
    """-> (rows buffer (B*H*W,4) int64, count (1,) int32 device)."""
    gV, mp, cc = _f32c(gV), _f32c(map_points), _f32c(map_ccounts)
    B, _, H, W = gV.shape[:4]
    dev = gV.device
    out = torch.empty((B * H * W, 4), dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = workspace(ws_bytes("gs_fusion_unique_ws_bytes", B, H, W), dev, "unique")
    call("gs_fusion_unique", ptr(rows), ptr(keep), ptr(n_rows_dev), max_rows, ptr(gV), B, H, W, ptr(mp), ptr(cc),
         mp.shape[1], ptr(out), ptr(cnt), ptr(ws), ws.numel(), stream())
    return out, cnt


def fusion_merge_raw(rows, n_rows_dev, max_rows, gV, gN, rgb, alpha, counts_dev, mp, mn, mc, cc):
    """-> new (points, normals, colors, ccounts), each a fresh (B,Nmax,C) tensor."""
    B, _, H, W = gV.shape[:4]
    Nmax = mp.shape[1]
    dev = gV.device
    op, on, oc, occ = (torch.empty_like(x) for x in (mp, mn, mc, cc))
    ws = workspace(ws_bytes("gs_fusion_merge_ws_bytes", B, Nmax), dev, "merge")
    call("gs_fusion_merge", ptr(rows), ptr(n_rows_dev), max_rows, ptr(gV), ptr(gN), ptr(rgb), ptr(alpha), B, H, W, Nmax,
         ptr(counts_dev), ptr(mp), ptr(mn), ptr(mc), ptr(cc), ptr(op), ptr(on), ptr(oc), ptr(occ), ptr(ws), ws.numel(),
         stream())
    return op, on, oc, occ


class _MergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, counts_dev, gV, gN, rgb, alpha, mp, mn, mc, cc):
        gV, gN, rgb, alpha, mp, mn, mc, cc = (_f32c(x) for x in (gV, gN, rgb, alpha, mp, mn, mc, cc))
        n_dev = dev_int(rows.shape[0], rows.device)
        outs = fusion_merge_raw(rows, n_dev, rows.shape[0], gV, gN, rgb, alpha, counts_dev, mp, mn, mc, cc)
        ctx.save_for_backward(rows, counts_dev, gV, gN, rgb, alpha, mp, mn, mc, cc)
        return outs

    @staticmethod
    def backward(ctx, gop, gon, goc, gocc):
        rows, counts_dev, gV, gN, rgb, alpha, mp, mn, mc, cc = ctx.saved_tensors
        B, _, H, W = gV.shape[:4]
        Nmax = mp.shape[1]
        dev = gV.device
        gop, gon, goc, gocc = (None if g is None else _f32c(g) for g in (gop, gon, goc, gocc))
        gip, ginn, gic, gicc = (torch.empty_like(x) for x in (mp, mn, mc, cc))
        ggv, ggn, grgb, galpha = (torch.zeros_like(x) for x in (gV, gN, rgb, alpha))
        ws = workspace(ws_bytes("gs_fusion_merge_ws_bytes", B, Nmax), dev, "merge")
        call("gs_fusion_merge_backward", ptr(rows), ptr(dev_int(rows.shape[0], dev)), rows.shape[0], ptr(gV), ptr(gN),
             ptr(rgb), ptr(alpha), B, H, W, Nmax, ptr(counts_dev), ptr(mp), ptr(mn), ptr(mc), ptr(cc), ptr(gop), ptr(gon),
             ptr(goc), ptr(gocc), ptr(gip), ptr(ginn), ptr(gic), ptr(gicc), ptr(ggv), ptr(ggn), ptr(grgb), ptr(galpha),
             ptr(ws), ws.numel(), stream())
        return None, None, ggv, ggn, grgb, galpha, gip, ginn, gic, gicc


def fusion_merge(rows, counts_dev, gV, gN, rgb, alpha, mp, mn, mc, cc):
    """Autograd-aware merge of the unique rows into the padded map arrays."""
    return _MergeFn.apply(rows, counts_dev, gV, gN, rgb, alpha, mp, mn, mc, cc)


def fusion_new_mask_raw(depth, rows, n_rows_dev, max_rows):
    """depth (B,1,H,W,1) -> mask (B,H,W) uint8: valid depth and not matched."""
    depth = _f32c(depth)
    B, _, H, W = depth.shape[:4]
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=depth.device)
    call("gs_fusion_new_mask", ptr(depth), ptr(rows), ptr(n_rows_dev), max_rows, B, H, W, ptr(mask), stream())
    return mask
