#!/usr/bin/env python3
"""Host-side cost of one c2 localisation step (enqueue only, no synchronisation) vs its GPU time."""
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
    for mode in (1, 0):
        _native.lib().gs_set_graph_mode(mode)
        for i in range(10):
            bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
        torch.cuda.synchronize()
        # GPU-bound rate
        t0 = time.perf_counter()
        for i in range(100):
            bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
        torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 100
        # host cost: enqueue 20 steps at a time into an idle queue, measure enqueue time only
        enq = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20):
                bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
            enq.append((time.perf_counter() - t0) / 20)
            torch.cuda.synchronize()
        print("graph" if mode else "eager", "step %.1f us | host enqueue %.1f us/step (min %.1f)" % (1e6 * t_all, 1e6 * sum(enq) / len(enq), 1e6 * min(enq)))
import cProfile, pstats
_native.lib().gs_set_graph_mode(1)
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for i in range(200):
        bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
