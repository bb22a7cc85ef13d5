#!/usr/bin/env python3
"""Diagnostic only: per-wave phase stamps of the pruned nearest-neighbour kernel on the c2 clouds.
Uses libgradslam_hip_diag.so (make -C gradslam_amd/csrc diag); prints where a block's time goes."""
import ctypes, math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gradslam_amd import _native
_native.LIB_PATH = os.path.join(ROOT, "gradslam_amd", "libgradslam_hip_diag.so")
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd.synthetic import make_sequence

dev = "cuda:0"
c, d, K, P = make_sequence(1, 2, 480, 640, seed=0)
fr = gs.RGBDImages(c.to(dev), d.to(dev), K.to(dev), P.to(dev))
slam = gs.slam.PointFusion(odom="icp", dsratio=4, numiters=10, device=dev)
with torch.no_grad():
    pcs, _ = slam.step(gs.Pointclouds(device=dev), fr[:, 0], None)
    live = gs.RGBDImages(c[:, 1:2].to(dev), d[:, 1:2].to(dev), K.to(dev), P[:, :1].to(dev))
    src = gs.odometry.icputils.downsample_rgbdimages(live, 4).points_list[0].contiguous()
    rows, cnt = gs.slam.fusionutils._project(pcs, fr[:, 0], 4)
    tgt = gs.odometry.icputils._gather_by_table(pcs, rows[: int(cnt.item())], 1).points_list[0].contiguous()
print("src", src.shape, "tgt", tgt.shape)
lib = _native.lib()
nblk = 2048  # >= the association's grid for any cloud these diagnostics use (mixed 64 / 16-point tiling)
dbg = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device=dev)
lib.gs_diag_set_buffer.argtypes = [ctypes.c_void_p]
for it in range(3):
    dbg.zero_()
    assert lib.gs_diag_set_buffer(dbg.data_ptr()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); best = ops.knn1_raw(src, tgt); e1.record(); torch.cuda.synchronize()
    print("call ms (incl. boxes kernel)", e0.elapsed_time(e1))
def report(title):
    global a
    print("====", title)
    a = (dbg.cpu().numpy().reshape(nblk, 16, 16) & np.array([-1, -1, -1, -1, -1, 0xffff] + [-1] * 10, dtype=np.int64)).astype(np.float64)
    tick = 1e-2
    t0, t1, t2, t3, ns = a[..., 0], a[..., 1], a[..., 2], a[..., 3], a[..., 4]
    g0 = t0.min()
    print("seed phase us  (per wave)  p50 %.2f p99 %.2f" % tuple(np.percentile((t1 - t0) * tick, [50, 99])))
    print("main loop us   (per wave)  p50 %.2f p99 %.2f max %.2f" % tuple(np.percentile((t2 - t1) * tick, [50, 99, 100])))
    print("block total us             p50 %.2f p99 %.2f max %.2f" % tuple(np.percentile((t3.max(1) - t0.min(1)) * tick, [50, 99, 100])))
    print("kernel span us %.1f" % ((t3.max() - g0) * tick))
    print("chunks scanned per block: mean %.1f max %d ; coarse survivors per block mean %.1f" % (ns.sum(1).mean(), ns.sum(1).max(), a[..., 5].max(1).mean()))


report("stand-alone search (sampled seed)")
nrm = gs.odometry.icputils._gather_by_table(pcs, rows[: int(cnt.item())], 1).normals_list[0].contiguous()
dbg.zero_()
T, _, _ = ops.icp_device_loop(src, tgt, nrm, torch.eye(4, device=dev), 10, 1e-8, None)
torch.cuda.synchronize()
report("last association of a 10-iteration ICP loop (seeded by the previous neighbour)")
a = (dbg.cpu().numpy().reshape(nblk, 16, 16) & np.array([-1, -1, -1, -1, -1, 0xffff] + [-1] * 10, dtype=np.int64)).astype(np.float64)
t6, t7, t0, t3 = a[..., 6], a[..., 7], a[..., 0], a[..., 3]
print("folded step (prologue) us per wave: p50 %.2f p99 %.2f ; prologue end -> search start p50 %.2f ; search p50 %.2f ; kernel entry spread p99 %.2f" % (
    *np.percentile((t7 - t6) * 0.01, [50, 99]), np.percentile((t0 - t7) * 0.01, 50), np.percentile((t3 - t0) * 0.01, 50),
    np.percentile((t6 - t6.min()) * 0.01, 99)))
print("kernel span incl. prologue us %.1f" % ((t3.max() - t6.min()) * 0.01))
# step-kernel stamps live in slots 0..4 of the buffer; rerun the loop with a separate small buffer
dbg2 = torch.zeros(8, dtype=torch.int64, device=dev)
assert lib.gs_diag_set_buffer(dbg2.data_ptr()) == 0
# (the association kernels also write into this buffer at other offsets: give them room)
dbg3 = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device=dev)
assert lib.gs_diag_set_buffer(dbg3.data_ptr()) == 0
T, _, _ = ops.icp_device_loop(src, tgt, nrm, torch.eye(4, device=dev), 10, 1e-8, None)
torch.cuda.synchronize()
st = dbg3[:8].cpu().numpy().astype(np.float64)
print("==== last icp_step_k (us): copy-in+reduce %.2f | decide %.2f | solve6 %.2f | rest(exp, out) %.2f | total %.2f" % (
    (st[1] - st[0]) * 0.01, (st[3] - st[1]) * 0.01, (st[4] - st[3]) * 0.01, (st[2] - st[4]) * 0.01, (st[2] - st[0]) * 0.01))
a = (dbg.cpu().numpy().reshape(nblk, 16, 16) & np.array([-1, -1, -1, -1, -1, 0xffff] + [-1] * 10, dtype=np.int64)).astype(np.float64)
tick = 1e-2  # wall_clock64: 100 MHz -> 10 ns per tick = 0.01 us
t0, t1, t2, t3, ns = a[..., 0], a[..., 1], a[..., 2], a[..., 3], a[..., 4]
g0 = t0.min()
print("block start spread us: p50 %.1f p99 %.1f max %.1f" % tuple(np.percentile((t0.min(1) - g0) * tick, [50, 99, 100])))
print("seed phase us  (per wave)  p50 %.2f p99 %.2f" % tuple(np.percentile((t1 - t0) * tick, [50, 99])))
print("main loop us   (per wave)  p50 %.2f p99 %.2f max %.2f" % tuple(np.percentile((t2 - t1) * tick, [50, 99, 100])))
print("barrier wait us(per wave)  p50 %.2f p99 %.2f" % tuple(np.percentile((t3 - t2) * tick, [50, 99])))
print("block total us             p50 %.2f p99 %.2f max %.2f" % tuple(np.percentile((t3.max(1) - t0.min(1)) * tick, [50, 99, 100])))
print("kernel span us %.1f" % ((t3.max() - g0) * tick))
print("chunks scanned per wave: mean %.2f max %d ; per block: mean %.1f max %d ; coarse survivors per block mean %.1f" % (ns.mean(), ns.max(), ns.sum(1).mean(), ns.sum(1).max(), a[..., 5].max(1).mean()))
