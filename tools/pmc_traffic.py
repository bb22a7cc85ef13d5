#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (wrap in `rocprofv3 --pmc FETCH_SIZE ...` and, separately,
`--pmc WRITE_SIZE`): (a) calibration: gs_transform_points on 2^24 points = exactly 12 B read + 12 B written
per point with the same 3 x dword-per-lane access pattern the J kernel uses; (b) gs_icp_linearize on 2^24
points with image-coherent associations (the `roofline_hbm` workload of bench.py)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd._native import call, ptr, stream, workspace, ws_bytes

dev = "cuda:0"
n = 1 << 24
g = torch.Generator(device=dev).manual_seed(0)
src = torch.rand((n, 3), device=dev, generator=g)
tgt = src + 0.01
nrm = torch.nn.functional.normalize(torch.rand((n, 3), device=dev, generator=g), dim=-1)
idx = (torch.arange(n, device=dev) + torch.randint(-8, 9, (n,), device=dev, generator=g)).clamp_(0, n - 1)
best = (torch.full((n,), 1e-4, device=dev).view(torch.int32).to(torch.int64) << 32) | idx
out = torch.empty(44, device=dev)
ws = workspace(ws_bytes("gs_icp_linearize_ws_bytes", n), dev, "linearize")
d_n = ops.dev_int(n, dev)
T = torch.eye(4, device=dev)
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(5):
    call("gs_transform_points", ptr(src), ptr(d_n), n, ptr(T), ptr(dst), stream())
for _ in range(5):
    call("gs_icp_linearize", ptr(src), ptr(d_n), n, ptr(tgt), ptr(nrm), ptr(best), -1.0, ptr(out), ptr(ws), ws.numel(), stream())
torch.cuda.synchronize()
print("done", n)
