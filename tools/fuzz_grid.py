#!/usr/bin/env python3
"""One-off wider fuzz of the grid search with distance certificates (the permanent test runs 8 seeds):
python tools/fuzz_grid.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gradslam_amd as gs
from tests import test_gpu_parity as T

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 160)
bad = 0
for seed in range(first, first + count):
    try:
        T.test_grid_search_fuzz_against_bruteforce.__wrapped__(gs, seed) if hasattr(T.test_grid_search_fuzz_against_bruteforce, "__wrapped__") else \
            T.test_grid_search_fuzz_against_bruteforce(gs, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:300], flush=True)
print("seeds", count, "failures", bad)
