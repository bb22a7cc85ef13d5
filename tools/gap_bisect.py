#!/usr/bin/env python3
"""Where do the clean run's gaps come from?  The 200-frame PointFusion forward spends ~0.39 ms per frame in kernels (and runs
at that rate under rocprofv3) but ~0.61 ms per frame clean, with the host 4x ahead; the c2 step alone shows no such gap.
This replays ICPSLAM._forward_streamed by hand with its per-frame host-visible operations switched off one at a time
(exact row bounds from a first pass, so that nothing else changes):
    readback  the asynchronous D2H copy of the map counts + the event behind it (arena.appended)
    posecopy  recovered[:, s] = pose[:, 0]
    stats     the per-frame statistics row
usage: gap_bisect.py [frames=200]"""
import math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
H, W = 480, 640
c, d, K, P = (x.to(dev) for x in make_sequence(1, n, H, W, seed=100))
slam = gs.slam.PointFusion(odom="icp", dsratio=4, numiters=10, device=dev)
with torch.no_grad():
    pcs, poses_ref = slam(gs.RGBDImages(c, d, K, P))
counts = torch.tensor(slam.last_appended).sum(1).cumsum(0).tolist()  # map size after every frame
cap = 1 << (counts[-1] + H * W).bit_length()
p = slam.odomprov
dot_th = slam.dot_th


def run(readback=True, posecopy=True, stats_on=True, localize=True, update=True):
    mk = lambda w_: torch.zeros((1, cap, w_), device=dev)
    mp, mn, mc, mf = mk(3), mk(3), mk(3), mk(1)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    recovered = torch.empty((1, n, 4, 4), device=dev)
    stats = torch.zeros((n, 5), dtype=torch.int32, device=dev)
    pinned = [torch.empty(1, dtype=torch.int32).pin_memory() for _ in range(4)]
    pending = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prev = None
    for s in range(n):
        d_s, c_s = d[:, s], c[:, s]
        have = counts[s - 1] if s else 0
        if s == 0 or not localize:
            pose = P[:, s:s + 1]
        else:
            pose, _, _ = ops.slam_localize_raw(d_s.unsqueeze(1), K, prev, mp[:, :have], mn[:, :have], cnt, 4, p.numiters, p.damp, p.dist_thresh, None)
        bound = have + H * W
        if update:
            ops.pointfusion_update_raw(d_s, c_s, K, pose, mp[:, :bound], mn[:, :bound], mc[:, :bound], mf[:, :bound], cnt, slam.dist_th, dot_th,
                                       slam.sigma, stats[s] if stats_on else None)
        if readback:
            while pending and pending[0][1].query():
                pinned.append(pending.pop(0)[0])
            if pinned:
                buf = pinned.pop()
                buf.copy_(cnt, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending.append((buf, ev))
        if posecopy:
            recovered[:, s] = pose[:, 0]
        prev = pose
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return 1e3 * dt / n, 1e3 * t_enq / n, int(cnt.item())


with torch.no_grad():
    run()
    for label, kw in (("everything", {}), ("no readback", dict(readback=False)), ("no pose copy", dict(posecopy=False)),
                      ("no stats row", dict(stats_on=False)), ("none of the three", dict(readback=False, posecopy=False, stats_on=False)),
                      ("localisation only (gt-posed map update skipped)", dict(update=False, readback=False, posecopy=False)),
                      ("map update only (ground-truth poses)", dict(localize=False, readback=False, posecopy=False)),
                      ("everything", {})):
        ms, enq, m = run(**kw)
        print("%-52s %.4f ms/frame (%.0f frames/s), host enqueue %.4f ms/frame, map %d" % (label, ms, 1e3 / ms, enq, m))
