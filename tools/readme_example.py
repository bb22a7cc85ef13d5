#!/usr/bin/env python3
"""Runs the usage snippet of README.md as written (needs an MI355X)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
text = open(os.path.join(ROOT, "README.md")).read()
code = re.search(r"```python\n(.*?)```", text, re.S).group(1)
scope = {}
exec(compile(code, "README.md", "exec"), scope)
pcs, poses = scope["pointclouds"], scope["live_poses"]
print("map points", pcs.num_points_per_pointcloud.tolist(), "recovered", tuple(scope["recovered_poses"].shape), "live", tuple(poses.shape))
