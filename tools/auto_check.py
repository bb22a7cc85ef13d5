#!/usr/bin/env python3
"""c2 step in the library's automatic launch mode: time per block of steps (does it settle, and where)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import gradslam_amd as gs
from gradslam_amd import _native
dev = torch.device("cuda", 0)
slam, world_map, prev, lives, K, raw = bench.build_workload(gs, dev, seed=0)
with torch.no_grad():
    for mode in (-1, 1, 0, -1):
        _native.lib().gs_set_graph_mode(mode)
        out = []
        for blk in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(25):
                bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
            torch.cuda.synchronize(); out.append(1e6 * (time.perf_counter() - t0) / 25)
        print("mode", mode, " ".join("%.0f" % x for x in out))
