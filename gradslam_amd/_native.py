"""ctypes binding of libgradslam_hip.so (the C ABI declared in include/gradslam_hip.h).

PyTorch is used only as plumbing: it owns the HBM allocations (`tensor.data_ptr()`) and the HIP
stream (`torch.cuda.current_stream().cuda_stream`); every kernel on the hot path lives in the
shared library.  There is NO CPU fallback: if the library is missing, or a tensor is not on a HIP
device, the call raises.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgradslam_hip.so")
CSRC = os.path.join(_HERE, "csrc")

c_i, c_i64, c_f, c_p, c_sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, [argtypes]); mirrors include/gradslam_hip.h one to one
SIGNATURES = {
    "gs_abi_version": (c_i, []),
    "gs_last_error": (ctypes.c_char_p, []),
    "gs_vertex_normal_maps": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "gs_vertex_normal_maps_backward_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "gs_vertex_normal_maps_backward": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                             c_p, c_sz, c_p]),
    "gs_get_alpha": (c_i, [c_p, c_i64, c_f, c_f, c_p, c_p]),
    "gs_get_alpha_backward": (c_i, [c_p, c_i64, c_f, c_f, c_p, c_p, c_p]),
    "gs_frames_from_raw": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p, c_p]),
    "gs_compact_ws_bytes": (c_sz, [c_i64]),
    "gs_compact_rows": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gs_compact_multi": (c_i, [c_i, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_sz, c_p]),
    "gs_expand_multi": (c_i, [c_i, c_p, c_p, c_p, c_p, c_i64, c_p, c_sz, c_p]),
    "gs_append_rows_ws_bytes": (c_sz, [c_i64]),
    "gs_append_rows": (c_i, [c_i, c_p, c_p, c_p, c_p, c_i64, c_p, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gs_fusion_merge_inplace_ws_bytes": (c_sz, [c_i, c_i]),
    "gs_fusion_merge_inplace": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p,
                                      c_sz, c_p]),
    "gs_aggregate_update_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "gs_aggregate_update": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_sz, c_p]),
    "gs_pointfusion_update_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "gs_pointfusion_update": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_p, c_p,
                                    c_sz, c_p]),
    "gs_downsample_frame_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "gs_downsample_frame": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gs_build_icp_target_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "gs_build_icp_target": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                  c_p, c_p, c_sz, c_p]),
    "gs_bucket_by_pixel_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "gs_bucket_by_pixel": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gs_project_active_ws_bytes": (c_sz, [c_i, c_i]),
    "gs_project_active": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gs_gather_table_rows_ws_bytes": (c_sz, [c_i]),
    "gs_gather_table_rows": (c_i, [c_p, c_p, c_i64, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gs_table_ds_mask": (c_i, [c_p, c_i64, c_i, c_p, c_p]),
    "gs_knn1_ws_bytes": (c_sz, [c_i]),
    "gs_knn1": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_sz, c_p]),
    "gs_knn1_bruteforce": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p]),
    "gs_knn1_unpack": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p]),
    "gs_icp_linearize_ws_bytes": (c_sz, [c_i]),
    "gs_icp_linearize": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_sz, c_p]),
    "gs_icp_rows": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p]),
    "gs_icp_linearize_backward": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p]),
    "gs_transform_points": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p]),
    "gs_icp_ws_bytes": (c_sz, [c_i, c_i]),
    "gs_icp_point_to_plane": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_sz,
                                    c_p]),
    "gs_icp_point_to_plane_grad": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_f,
                                         c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gs_icp_tape_bytes": (c_sz, [c_i, c_i, c_i]),
    "gs_icp_point_to_plane_taped": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f,
                                          c_p, c_p, c_p, c_p, c_sz, c_p, c_sz, c_p]),
    "gs_icp_backward_ws_bytes": (c_sz, [c_i]),
    "gs_icp_point_to_plane_backward": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_p, c_sz,
                                             c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gs_slam_localize_tape_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "gs_slam_localize_taped": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_f,
                                     c_f, c_f, c_p, c_p, c_sz, c_p, c_sz, c_p]),
    "gs_slam_localize_backward_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "gs_slam_localize_backward": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_p, c_sz,
                                        c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_sz, c_p]),
    "gs_pointfusion_update_tape_bytes": (c_sz, [c_i, c_i, c_i]),
    "gs_pointfusion_update_taped": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_p, c_p,
                                          c_sz, c_p, c_sz, c_p]),
    "gs_pointfusion_update_backward_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "gs_pointfusion_update_backward": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_p, c_sz,
                                             c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gs_compose_poses": (c_i, [c_p, c_p, c_i, c_p, c_p]),
    "gs_set_graph_mode": (None, [c_i]),
    "gs_graph_stats": (c_i, [ctypes.POINTER(ctypes.c_double)]),
    "gs_slam_localize_ws_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "gs_slam_localize": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f,
                               c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gs_set_grid_search": (None, [c_i]),
    "gs_set_tile_points": (None, [c_i]),
    "gs_icp_launch_geometry": (c_i, [c_i, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i), ctypes.POINTER(c_i)]),
    "gs_loop_counts": (c_i, [ctypes.POINTER(ctypes.c_uint), c_i]),
    "gs_profile_enable": (None, [c_i]),
    "gs_profile_read": (c_i, [c_i, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double)]),
    "gs_fusion_similar": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_f, c_f, c_p, c_p, c_p]),
    "gs_fusion_unique_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "gs_fusion_unique": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_sz, c_p]),
    "gs_fusion_merge_ws_bytes": (c_sz, [c_i, c_i]),
    "gs_fusion_merge": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p,
                              c_p, c_p, c_p, c_p, c_sz, c_p]),
    "gs_fusion_merge_backward": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p,
                                       c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_sz,
                                       c_p]),
    "gs_fusion_new_mask": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", CSRC, "-j4"]
    if verbose:
        cmd.remove("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "gradslam_amd: native library {} is missing -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU / PyTorch fallback for the hot path.".format(LIB_PATH)
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        if handle.gs_abi_version() != 3:
            raise RuntimeError("gradslam_amd: ABI version mismatch")
        _lib = handle
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """Raw handle of torch's current HIP stream on the current device.  The raw accessors cost well under a
    microsecond; torch.cuda.current_stream() builds a Stream object (~10 us), which matters at ~80 us of host
    time per localisation step."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def current_device_index() -> int:
    """Index of the HIP device whose current stream `stream()` hands to the kernels."""
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def require_hip(*tensors, op: str = "op"):
    """No CPU fallback: the hot path only runs on HIP tensors -- and only on the CURRENT device.  The C side
    takes raw pointers and a stream and never calls hipSetDevice: launching on the current device's stream with
    another device's pointers would fault, so a tensor elsewhere is an error here (one process per GPU, device
    selected with torch.cuda.set_device, is the deployment model; parallel.init_from_env does that)."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "gradslam_amd.{}: tensor is on {}; the ICP / PointFusion hot path only runs on a HIP "
                "device (no CPU fallback is provided).".format(op, t.device)
            )
        if cur is None:
            cur = current_device_index()
        if t.device.index != cur:
            raise RuntimeError(
                "gradslam_amd.{}: tensor is on {} but the current HIP device is cuda:{}; kernels are launched on "
                "the current device's stream -- call torch.cuda.set_device({}) (or wrap the call in "
                "`with torch.cuda.device(...)`) first.".format(op, t.device, cur, t.device.index)
            )


def call(name: str, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        msg = lib().gs_last_error().decode("utf-8", "replace")
        raise RuntimeError("{} failed (code {}): {}".format(name, rc, msg))


def ws_bytes(name: str, *args) -> int:
    return int(getattr(lib(), name)(*args))


_ws_cache = {}


def workspace(nbytes: int, device, tag: str = "default") -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream, tag).  Kernels of one call finish with it before
    the next call on the same stream can start, so reuse is safe.  Sizes are rounded up to a power of two:
    a map that grows frame by frame then keeps its workspace ADDRESS for many frames, which is what lets
    gs_slam_localize replay its captured graph (the address is part of the graph's key)."""
    key = (str(device), stream(), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(1 << max(int(nbytes) - 1, 255).bit_length(), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
