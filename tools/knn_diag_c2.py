#!/usr/bin/env python3
"""Diagnostic only: phase stamps of the LAST association launch of the bench's c2 localisation step when the loop is
cut after k iterations (k = 0 .. 4): what the 1st, 2nd, ... association of a loop costs and how many lanes need the
exact search there (needs make -C gradslam_amd/csrc diag).  usage: knn_diag_c2.py [live frame index 1..4]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gradslam_amd import _native
_native.LIB_PATH = os.path.join(ROOT, "gradslam_amd", "libgradslam_hip_diag.so")
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence

live_idx = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
c, d, K, P = make_sequence(1, 5, 480, 640, seed=0)
c, d, K, P = c.to(dev), d.to(dev), K.to(dev), P.to(dev)
frames = gs.RGBDImages(c, d, K, P)
lib = _native.lib()
lib.gs_diag_set_buffer.argtypes = [ctypes.c_void_p]
nblk = 2048
dbg = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device=dev)
assert lib.gs_diag_set_buffer(dbg.data_ptr()) == 0
tick = 1e-2
for k in range(0, 6):
    slam = gs.slam.PointFusion(odom="icp", dsratio=4, numiters=max(k, 1), device=dev)
    with torch.no_grad():
        world_map, _ = slam.step(gs.Pointclouds(device=dev), frames[:, 0], None)
        if k == 0:
            continue
        dbg.zero_()
        live = gs.RGBDImages(c[:, live_idx:live_idx + 1].contiguous(), d[:, live_idx:live_idx + 1].contiguous(), K)
        slam._localize(world_map, live, frames[:, 0])
    torch.cuda.synchronize()
    raw = dbg.cpu().numpy().reshape(nblk, 16, 16)
    a = raw.astype(np.float64)
    live_b = a[..., 3].max(1) > 0
    a = a[live_b]
    t0, t1, t2, t3, t6, t7 = a[..., 0], a[..., 1], a[..., 2], a[..., 3], a[..., 6], a[..., 7]
    need = a[:, 0, 12]
    p = lambda x: np.percentile(x, 50)
    print("association %d of the loop (live frame %d): prologue %.2f | seed+window %.2f | verify %.2f | block total p50 %.2f max %.2f | span %.1f us | "
          "lanes needing exact search per tile mean %.1f, tiles with none %d/%d | coarse %.2f fine %.2f barrier %.2f (p50 per wave, where run) scanned/tile %.1f" % (
              k + 1, live_idx, p((t7 - t6) * tick), p((t1 - t0) * tick), p((t2 - t1) * tick), p((t3.max(1) - t6.min(1)) * tick),
              ((t3.max(1) - t6.min(1)) * tick).max(), (t3.max() - t6.min()) * tick, need.mean(), int((need == 0).sum()), need.shape[0],
              p(a[..., 8] * tick), p(a[..., 9] * tick), p(a[..., 10] * tick), a[..., 4].sum(1).mean()))
    why = raw[live_b][:, 0, 13]
    rad = raw[live_b][:, 0, 14]
    f = lambda sh: ((why >> sh) & 0xff).mean()
    print("     lanes per tile without a proof because: window not fully staged %.1f | smaller radius than the certificate's %.1f | no certificate (m = 0) %.1f | "
          "moved beyond its radius %.1f | window's best not inside the reach %.1f ; window radius now %.2f ; tile means (mm): sqrt(m) %.1f moved %.1f sqrt(bd) %.1f" % (
              f(0), f(8), f(16), f(24), f(32), ((why >> 40) & 0xff).mean(), (rad & 0xfffff).mean() / 1e3, ((rad >> 20) & 0xfffff).mean() / 1e3,
              ((rad >> 40) & 0xfffff).mean() / 1e3))
