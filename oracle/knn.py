"""ctypes front for oracle/knn_ref.c (test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libknn_ref.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "_build/libknn_ref.so"])


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "knn_ref.c")
        ):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.knn1_ref.restype = None
        _lib.knn1_ref_f64.restype = None
        _lib.knn1_ref_wide.restype = None
    return _lib


WIDE = False  # tests of long 640x480 sequences switch the sixteen-points-per-pass form on


def knn1(src: torch.Tensor, tgt: torch.Tensor, wide=None):
    """src (Ns,3), tgt (Nt,3) fp32 CPU -> (dist2 (Ns,) fp32, idx (Ns,) int64).
    wide=True runs knn1_ref_wide (sixteen points per pass, bit-identical, ~10x faster); None = the module's WIDE."""
    wide = WIDE if wide is None else wide
    lib = _load()
    s = np.ascontiguousarray(src.detach().cpu().numpy(), dtype=np.float32)
    t = np.ascontiguousarray(tgt.detach().cpu().numpy(), dtype=np.float32)
    d = np.empty(s.shape[0], dtype=np.float32)
    i = np.zeros(s.shape[0], dtype=np.int64)
    if s.shape[0] and t.shape[0]:
        (lib.knn1_ref_wide if wide else lib.knn1_ref)(s.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(s.shape[0]),
                     t.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(t.shape[0]),
                     d.ctypes.data_as(ctypes.c_void_p), i.ctypes.data_as(ctypes.c_void_p))
    return torch.from_numpy(d), torch.from_numpy(i)


def knn1_f64(src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    lib = _load()
    s = np.ascontiguousarray(src.detach().cpu().numpy(), dtype=np.float32)
    t = np.ascontiguousarray(tgt.detach().cpu().numpy(), dtype=np.float32)
    d = np.empty(s.shape[0], dtype=np.float64)
    lib.knn1_ref_f64(s.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(s.shape[0]),
                     t.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(t.shape[0]),
                     d.ctypes.data_as(ctypes.c_void_p))
    return torch.from_numpy(d)


def set_threads(n: int):
    """OMP threads for the C search (cpu_baseline reports what it used)."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass
