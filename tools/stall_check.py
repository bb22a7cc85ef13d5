#!/usr/bin/env python3
"""Does the host stall when it runs far ahead of the GPU?  Per-step host times over 300 steps, free-running vs
with an event fence that keeps at most DEPTH steps in flight."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import gradslam_amd as gs
dev = torch.device("cuda", 0)
slam, world_map, prev, lives, K, raw = bench.build_workload(gs, dev, seed=0)
def run(depth, n=300):
    evs = []
    torch.cuda.synchronize(); t0 = time.perf_counter(); last = t0; worst = []
    for i in range(n):
        bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
        if depth:
            e = torch.cuda.Event(); e.record(); evs.append(e)
            if len(evs) > depth:
                evs.pop(0).synchronize()
        now = time.perf_counter(); worst.append(now - last); last = now
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    worst_sorted = sorted(worst, reverse=True)[:4]
    print("depth %s: %.1f us/step overall; largest host step times (ms): %s at %s" % (
        depth or "free", 1e6 * dt / n, " ".join("%.2f" % (1e3 * w) for w in worst_sorted),
        [worst.index(w) for w in worst_sorted]))
with torch.no_grad():
    for d in (0, 16, 0, 4):
        run(d)
