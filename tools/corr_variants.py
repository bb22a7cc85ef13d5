#!/usr/bin/env python3
"""What bounds the map-wide kernels of the PointFusion update?  Builds a real map (n frames of the c3 sequence through
the product), then times tools/micro/corr_variants.hip's switchable copies of corr_pass1_k / merge_corr_k on it.
Run it under `rocprofv3 --kernel-trace --stats` for exact per-kernel durations (the template arguments are in the
kernel names); the event timings printed here include ~2 us of launch per call.

usage: corr_variants.py [frames=100]"""
import ctypes, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libcorr_variants.so"))


class Args(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in
                ("mp", "mn", "cc", "counts", "poses", "Ks", "gv", "gn", "rgb", "alpha", "p", "nn", "cl", "ccw", "pix_key", "pix_n",
                 "pt_pix", "part_a", "part_s", "part_u")] + [("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                                                             ("dist_th", ctypes.c_float), ("dot_th", ctypes.c_float)]


lib.corr_variant.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Args), ctypes.c_void_p]
c, d, K, P = make_sequence(1, n + 1, 480, 640, seed=100)
H, W = 480, 640
slam = gs.slam.PointFusion(odom="icp", dsratio=4, numiters=10, device=dev)
with torch.no_grad():
    pcs, poses = slam(gs.RGBDImages(c[:, :n].to(dev), d[:, :n].to(dev), K.to(dev), P[:, :n].to(dev)))
    N = int(pcs.num_points_per_pointcloud.item())
    mp = pcs.points_padded[0, :N].contiguous()
    mn = pcs.normals_padded[0, :N].contiguous()
    mc = pcs.colors_padded[0, :N].contiguous()
    cc = pcs.features_padded[0, :N].reshape(-1).contiguous()
    pose = poses[:, n - 1].contiguous().float()  # the next frame seen from the last recovered pose: what the update of frame n sees, nearly
    K4 = K.to(dev).contiguous().float()
    Kd = K4[:, 0].contiguous()
    _, _, gv, gn = gs.ops.vertex_normal_maps_raw(d[:, n:n + 1].to(dev).contiguous(), K4, pose.reshape(1, 1, 4, 4), want_local=False,
                                                 want_global=True)
    rgb = c[:, n].to(dev).contiguous()
    alpha = (torch.rand(H * W, device=dev) * 0.9 + 0.1).contiguous()
print("map points", N, "gv", tuple(gv.shape), "gn", tuple(gn.shape))
counts = torch.tensor([N], dtype=torch.int32, device=dev)
KEY0 = torch.full((H * W,), -1, dtype=torch.int64, device=dev)
pix_key = KEY0.clone()
pix_n = torch.full((H * W,), -1, dtype=torch.int32, device=dev)
pt_pix = torch.empty(N + 4096, dtype=torch.int32, device=dev)
parts = [torch.zeros(N // 256 + 16, dtype=torch.int32, device=dev) for _ in range(3)]
wp, wn, wc, wcc = mp.clone(), mn.clone(), mc.clone(), cc.clone()
a = Args(mp.data_ptr(), mn.data_ptr(), cc.data_ptr(), counts.data_ptr(), pose.data_ptr(), Kd.data_ptr(), gv.data_ptr(), gn.data_ptr(),
         rgb.data_ptr(), alpha.data_ptr(), wp.data_ptr(), wn.data_ptr(), wc.data_ptr(), wcc.data_ptr(), pix_key.data_ptr(),
         pix_n.data_ptr(), pt_pix.data_ptr(), parts[0].data_ptr(), parts[1].data_ptr(), parts[2].data_ptr(), N, H, W, 0.05,
         math.cos(math.radians(20.0)))
st = torch.cuda.current_stream().cuda_stream


def run(which, I, F, cap, reps=6):
    ts = []
    for r in range(reps):
        if which == 0:
            pix_key.copy_(KEY0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.corr_variant(which, I, F, cap, ctypes.byref(a), st)
        e1.record()
        assert rc == 0, rc
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts[1:])[len(ts[1:]) // 2]


ref_key = ref_pix = None
print("== pass 1 (product: I=4 F=0 cap=0)")
for cap in (0, 2048):
    for I in (1, 2, 4, 8):
        for F in (0, 1, 3, 8, 24):
            if cap and F not in (0, 24):
                continue
            t = run(0, I, F, cap)
            note = ""
            if F in (0, 4, 8, 24):
                if ref_key is None:
                    ref_key, ref_pix = pix_key.clone(), pt_pix[:N].clone()
                    print("active", int(parts[0][: (N + 256 * I - 1) // (256 * I)].sum()), "similar",
                          int(parts[1][: (N + 256 * I - 1) // (256 * I)].sum()), "pixels with a candidate", int((ref_key != -1).sum()))
                else:
                    note = "same" if torch.equal(pix_key, ref_key) and torch.equal(pt_pix[:N], ref_pix) else "DIFFERENT"
            print(f"pass1 I={I} F={F} cap={cap:5d}  {t:7.1f} us {note}")
# a plausible winner table for the merge: pass 2 by torch (smallest index among the candidates of each pixel that hold the key is
# not needed for timing -- any ~one winner per pixel will do): take the first similar point of each pixel
pix_key.copy_(KEY0)
lib.corr_variant(0, 4, 0, 0, ctypes.byref(a), st)
sim = (pt_pix[:N] >= 0).nonzero().reshape(-1)
pix_n.fill_(-1)
pix_n.view(torch.int32).index_put_((pt_pix[:N][sim].long(),), sim.int(), accumulate=False)
print("== merge (product: I=4 F=0 cap=0); winners", int((pix_n != -1).sum()))
for cap in (0, 2048):
    for I in (1, 2, 4, 8):
        for F in (0, 1, 2):
            if cap and F != 2:
                continue
            wp.copy_(mp); wn.copy_(mn); wc.copy_(mc); wcc.copy_(cc)
            lib.corr_variant(1, I, F, cap, ctypes.byref(a), st)
            note = ""
            if F != 1:
                got = torch.cat([wp.reshape(-1), wn.reshape(-1), wc.reshape(-1), wcc])
                if I == 1 and F == 0 and cap == 0:
                    ref_m = got.clone()
                else:
                    note = "same" if torch.equal(got, ref_m) else "DIFFERENT"
            print(f"merge I={I} F={F} cap={cap:5d}  {run(1, I, F, cap):7.1f} us {note}")
