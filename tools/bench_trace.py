#!/usr/bin/env python3
"""bench.py's own timed loop with a time stamp every 5 steps (diagnostic for one-off costs inside the loop)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import gradslam_amd as gs
from gradslam_amd import parallel
rank, world, local = parallel.init_from_env()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t = time.perf_counter()
slam, world_map, prev, lives, K, raw = bench.build_workload(gs, dev, seed=0)
print("build_workload s %.2f" % (time.perf_counter() - t))
poses = []
with torch.no_grad():
    for i in range(5):
        bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); marks = []
    for i in range(50):
        poses.append(bench.one_step(gs, slam, world_map, prev, lives[i % 4], K))
        if i % 5 == 4:
            marks.append(time.perf_counter() - t0)   # host time, no sync
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("total ms %.3f  ms/step %.4f" % (1e3 * dt, 1e3 * dt / 50))
print("host marks (ms):", " ".join("%.2f" % (1e3 * m) for m in marks))
with torch.no_grad():
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(50):
            bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
        torch.cuda.synchronize()
        print("again: ms/step %.4f" % (1e3 * (time.perf_counter() - t0) / 50))
