"""CPU (no GPU) tests: the C ABI exports what include/gradslam_hip.h declares, the product refuses to
compute without a HIP device (no fallback), and the host-side mirror of the reference interface keeps
its container semantics and error contracts (messages matched like the reference's own tests do)."""
import os
import re

import pytest
import torch

import gradslam_amd as gs
from gradslam_amd import _native
from gradslam_amd.odometry import icputils
from gradslam_amd.slam import fusionutils

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI
def _declared():
    text = open(os.path.join(REPO, "include", "gradslam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.lib()  # loads without a GPU; no compute calls here
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_native.SIGNATURES) == names
    assert lib.gs_abi_version() == 3


def test_icp_launch_geometry_fits_the_workspace():
    """The loops' association launch (host-side rule, no device work): 64-point tiles unless a test forces another size;
    the workspace holds a partial row for every block under every tile-size setting."""
    import ctypes

    lib = _native.lib()

    def geom(max_ns, hints):
        b, t, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.gs_icp_launch_geometry(max_ns, hints, ctypes.byref(b), ctypes.byref(t), ctypes.byref(r)) == 0
        return b.value, t.value, r.value

    assert geom(19200, 1) == (300, 64, 601)          # 160 x 120 ds-grid (rows: 32-point tiles' worth + the spare row)
    assert geom(19200, 0)[:2] == (300, 64)
    assert geom(16385, 1)[:2] == (257, 64)
    assert geom(78408, 1)[:2] == (1226, 64)           # more blocks than the chip holds: unfolded steps
    try:
        for forced in (0, 32, 47, 64):
            lib.gs_set_tile_points(forced)
            for max_ns in (1, 63, 64, 65, 4800, 16385, 19200, 32768, 100000):
                for hints in (0, 1):
                    blocks, tile, rows = geom(max_ns, hints)
                    assert 32 <= tile <= 64 and blocks * tile >= max_ns and blocks < rows and rows >= 513, (forced, max_ns, hints)
                    if forced:
                        assert tile == forced
    finally:
        lib.gs_set_tile_points(0)
    assert lib.gs_icp_launch_geometry(0, 1, None, None, None) != 0  # error contract: positive capacity


def test_integration_notes_cover_every_entry_point():
    """INTEGRATION.md (the reference-side binding notes) names every symbol the header declares."""
    notes = open(os.path.join(REPO, "INTEGRATION.md")).read()
    header = open(os.path.join(REPO, "include", "gradslam_hip.h")).read()
    declared = set(re.findall(r"\b(gs_[a-z0-9_]+)\(", header))
    assert declared and not [n for n in sorted(declared) if n not in notes]


def test_integration_snippets_pass_the_declared_number_of_arguments():
    """Every `_lib.gs_*(...)` call in INTEGRATION.md's binding snippets has the arity of the bound signature."""
    from gradslam_amd import _native

    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = "\n".join(re.findall(r"```python\n(.*?)```", text, re.S))
    calls = 0
    for m in re.finditer(r"_lib\.(gs_[a-z0-9_]+)\(", code):
        name, j, depth, nargs, cur = m.group(1), m.end(), 1, 0, ""
        while depth > 0 and j < len(code):
            ch = code[j]
            depth += ch in "([{"
            depth -= ch in ")]}"
            if depth == 0:
                break
            if ch == "," and depth == 1:
                nargs, cur = nargs + 1, ""
            else:
                cur += ch
            j += 1
        nargs += bool(cur.strip())
        assert name in _native.SIGNATURES, name
        assert nargs == len(_native.SIGNATURES[name][1]), (name, nargs, len(_native.SIGNATURES[name][1]))
        calls += 1
    assert calls >= 15


def test_abi_is_plain_c():
    text = open(os.path.join(REPO, "include", "gradslam_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert 'extern "C"' in code
    assert "at::" not in code and "Tensor" not in code and "std::" not in code


def test_product_refuses_cpu_tensors():
    r = gs.RGBDImages(torch.rand(1, 1, 8, 8, 3), torch.rand(1, 1, 8, 8, 1), torch.eye(4).view(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.vertex_map
    pts = torch.rand(1, 10, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        icputils.point_to_plane_ICP(pts, pts, pts, torch.eye(4), numiters=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fusionutils.get_alpha(torch.rand(4, 3), 0.6)


def test_product_does_not_import_the_oracle():
    import subprocess
    import sys

    code = ("import sys, gradslam_amd; import gradslam_amd.slam, gradslam_amd.odometry; "
            "print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=REPO).decode().strip()
    assert out == "False"
    for root, _, files in os.walk(os.path.join(REPO, "gradslam_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


# ------------------------------------------------------------------ Pointclouds container
def _pcs():
    torch.manual_seed(0)
    pts = [torch.rand(5, 3), torch.rand(3, 3)]
    return gs.Pointclouds(pts, [p + 1 for p in pts], [p + 2 for p in pts], [torch.rand(5, 1), torch.rand(3, 1)]), pts


def test_pointclouds_list_padded_views():
    pc, pts = _pcs()
    assert len(pc) == 2 and not pc.equisized and pc.num_features == 1
    assert pc.points_padded.shape == (2, 5, 3) and torch.equal(pc.points_padded[1, :3], pts[1])
    assert (pc.points_padded[1, 3:] == 0).all()
    assert pc.nonpad_mask.tolist() == [[True] * 5, [True, True, True, False, False]]
    assert pc.num_points_per_pointcloud.tolist() == [5, 3]
    padded = gs.Pointclouds(torch.rand(2, 4, 3))
    assert padded.equisized and [p.shape[0] for p in padded.points_list] == [4, 4]
    assert not gs.Pointclouds().has_points and len(gs.Pointclouds()) == 0
    sub = pc[1]
    assert len(sub) == 1 and torch.equal(sub.points_list[0], pts[1])
    assert len(pc[[0, 1]]) == 2 and len(pc[torch.tensor([True, False])]) == 1
    with pytest.raises(IndexError):
        gs.Pointclouds()[0]


def test_pointclouds_append_and_setters():
    pc, pts = _pcs()
    other, _ = _pcs()
    pc.append_points(other)
    assert pc.num_points_per_pointcloud.tolist() == [10, 6] and pc.points_padded.shape == (2, 10, 3)
    assert torch.equal(pc.points_list[1][3:], pts[1]) and pc.features_padded.shape == (2, 10, 1)
    empty = gs.Pointclouds()
    empty.append_points(other)
    assert empty.has_points and empty.has_features and torch.equal(empty.points_list[0], pts[0])
    bad = pc.points_padded.clone()
    bad[1, -1] = 1.0
    with pytest.raises(ValueError, match="value must have zeros wherever"):
        pc.points_padded = bad
    good = pc.points_padded * 2
    pc.points_padded = good
    assert torch.equal(pc.points_list[0], good[0])
    with pytest.raises(ValueError, match="must either both have or not have normals"):
        pc.append_points(gs.Pointclouds([torch.rand(1, 3), torch.rand(1, 3)]))
    with pytest.raises(TypeError):
        pc.append_points(3)
    c = pc.clone()
    c.points_padded = c.points_padded * 0
    assert pc.points_padded.abs().sum() > 0
    assert (pc + 1.0).points_list[0].allclose(pc.points_list[0] + 1.0)
    assert pc.scale_(2.0).points_padded[1, -1].abs().sum() == 0


def test_pointclouds_constructor_contracts():
    with pytest.raises(TypeError, match="Expected points to be of type list or tensor or None"):
        gs.Pointclouds(3)
    with pytest.raises(TypeError, match="Expected normals to be of same type as points"):
        gs.Pointclouds([torch.rand(2, 3)], torch.rand(1, 2, 3))
    with pytest.raises(ValueError, match="last dim of all tensors in points should have shape 3"):
        gs.Pointclouds([torch.rand(2, 4)])
    with pytest.raises(ValueError, match="normals tensors should have same shape"):
        gs.Pointclouds([torch.rand(2, 3)], [torch.rand(3, 3)])
    with pytest.raises(ValueError, match=r"len\(points\) \(= 0\) should be > 0"):
        gs.Pointclouds([])
    with pytest.raises(ValueError, match="points should have ndim=3"):
        gs.Pointclouds(torch.rand(4, 3))


def test_pointclouds_transform_and_projection_algebra():
    pc, pts = _pcs()
    T = torch.eye(4)
    T[:3, 3] = torch.tensor([1.0, 2.0, 3.0])
    moved = pc.transform(T)
    assert moved.points_list[1].allclose(pts[1] + T[:3, 3]) and (moved.points_padded[1, 3:] == 0).all()
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 10.0
    proj = pc.pinhole_projection(K)
    want = torch.stack([10 * pts[0][:, 0] / pts[0][:, 2], 10 * pts[0][:, 1] / pts[0][:, 2], torch.ones(5)], -1)
    assert proj.points_list[0].allclose(want, rtol=1e-5)
    with pytest.raises(ValueError, match="transform should be of shape"):
        pc.transform(torch.eye(3))


# ------------------------------------------------------------------ RGBDImages container
def test_rgbdimages_contracts_and_indexing():
    rgb, depth = torch.rand(2, 3, 8, 6, 3), torch.rand(2, 3, 8, 6, 1)
    K = torch.eye(4).view(1, 1, 4, 4).repeat(2, 1, 1, 1)
    poses = torch.eye(4).view(1, 1, 4, 4).repeat(2, 3, 1, 1)
    r = gs.RGBDImages(rgb, depth, K, poses)
    assert r.shape == (2, 3, 8, 6) and len(r) == 2 and r.cdim == 4 and r.has_poses
    s = r[:, 1]
    assert s.shape == (2, 1, 8, 6) and torch.equal(s.poses, poses[:, 1:2])
    assert r[0].shape == (1, 3, 8, 6) and r[1, 0:2].shape == (1, 2, 8, 6)
    with pytest.raises(IndexError):
        r[5]
    with pytest.raises(IndexError):
        r[0, 0, 0]
    with pytest.raises(TypeError, match="Expected rgb_image to be of type tensor"):
        gs.RGBDImages(1, depth, K)
    with pytest.raises(ValueError, match="rgb_image should have ndim=5"):
        gs.RGBDImages(rgb[0], depth, K)
    with pytest.raises(ValueError, match="Expected depth_image to have shape"):
        gs.RGBDImages(rgb, depth[:, :2], K)
    with pytest.raises(ValueError, match="Expected intrinsics to have shape"):
        gs.RGBDImages(rgb, depth, K[:1])
    assert torch.equal(r.valid_depth_mask, depth > 0)
    cf = r.to_channels_first()
    assert cf.channels_first and cf.rgb_image.shape == (2, 3, 3, 8, 6)
    assert cf.to_channels_last().rgb_image.shape == rgb.shape
    r._vertex_map = torch.zeros(2, 3, 8, 6, 3)
    r._global_vertex_map = torch.zeros(2, 3, 8, 6, 3)
    r.poses = poses * 1
    assert r._global_vertex_map is None and r._vertex_map is not None
    r.intrinsics = K * 1
    assert r._vertex_map is None
    with pytest.raises(ValueError):
        r.poses = poses[:, :1]


# ------------------------------------------------------------------ error contracts of the path
def test_icputils_error_contracts():
    A, b = torch.rand(5, 6), torch.rand(5, 1)
    with pytest.raises(TypeError, match="Expected A to be of type torch.Tensor"):
        icputils.solve_linear_system(1, b)
    with pytest.raises(TypeError, match="Expected damp to be of type float or torch.Tensor"):
        icputils.solve_linear_system(A, b, 1)
    with pytest.raises(ValueError, match=r"b.shape\[1\] should 1"):
        icputils.solve_linear_system(A, torch.rand(5, 2))
    with pytest.raises(ValueError, match=r"A.shape\[0\] and b.shape\[0\] should be equal"):
        icputils.solve_linear_system(A, torch.rand(4, 1))
    A = torch.rand(50, 6)
    x = icputils.solve_linear_system(A, A @ torch.ones(6, 1), 1e-8)  # O(1) algebra: runs anywhere
    assert torch.allclose(x, torch.ones(6, 1), atol=1e-3)
    p = torch.rand(1, 7, 3)
    with pytest.raises(TypeError, match="Expected dist_thresh to be of type float or int"):
        icputils.gauss_newton_solve(p, p, p, "x")
    with pytest.raises(ValueError, match="src_pc should have ndim=3"):
        icputils.gauss_newton_solve(p[0], p, p)
    with pytest.raises(ValueError, match=r"tgt_pc.shape\[1\] and tgt_normals.shape\[1\] must be equal"):
        icputils.gauss_newton_solve(p, p, p[:, :5])
    with pytest.raises(TypeError, match="Expected numiters to be of type int"):
        icputils.point_to_plane_ICP(p, p, p, torch.eye(4), numiters=2.0)
    with pytest.raises(ValueError, match=r"Expected initial_transform.shape to be \(4, 4\)"):
        icputils.point_to_plane_ICP(p, p, p, torch.eye(3))
    with pytest.raises(TypeError, match="Expected lambda_max to be of type float or int"):
        icputils.point_to_plane_gradICP(p, p, p, torch.eye(4), lambda_max="2")
    pc = gs.Pointclouds([torch.rand(4, 3)])
    with pytest.raises(TypeError, match="Expected pointclouds to be of type gradslam.Pointclouds"):
        icputils.downsample_pointclouds(3, torch.zeros(1, 4, dtype=torch.int64), 2)
    with pytest.raises(ValueError, match=r"pc2im_bnhw.shape\[1\] must be 4"):
        icputils.downsample_pointclouds(pc, torch.zeros(1, 3, dtype=torch.int64), 2)
    with pytest.raises(TypeError, match="Expected ds_ratio to be of type int"):
        icputils.downsample_pointclouds(pc, torch.zeros(1, 4, dtype=torch.int64), 2.0)


def test_provider_and_slam_contracts():
    pts = torch.tensor([[5.0, 5.0, 5.0], [3.0, 3.0, 3.0]])
    for prov in (gs.odometry.ICPOdometryProvider(), gs.odometry.GradICPOdometryProvider()):
        with pytest.raises(ValueError, match="maps_pointclouds missing normals"):
            prov.provide(gs.Pointclouds([pts]), gs.Pointclouds([pts], [pts]))
        with pytest.raises(ValueError, match="Batch size of maps_pointclouds and frames_pointclouds should be equal"):
            prov.provide(gs.Pointclouds([pts], [pts]), gs.Pointclouds([pts, pts], [pts, pts]))
        with pytest.raises(TypeError, match="Expected maps_pointclouds to be of type gradslam.Pointclouds"):
            prov.provide(1, gs.Pointclouds([pts]))
    with pytest.raises(ValueError, match="not supported for PointFusion"):
        gs.slam.ICPSLAM(odom="xyz")
    with pytest.raises(TypeError, match="Distance threshold must be of type float or int"):
        gs.slam.PointFusion(dist_th="a")
    with pytest.warns(UserWarning, match="Angle threshold"):
        gs.slam.PointFusion(angle_th=120)
    slam = gs.slam.PointFusion(odom="icp")
    assert slam.dsratio == 4 and abs(slam.dot_th - 0.9396926) < 1e-6 and slam.sigma == 0.6
    with pytest.raises(TypeError, match="Expected frames to be of type gradslam.RGBDImages"):
        slam(3)
    r = gs.RGBDImages(torch.rand(1, 1, 4, 4, 3), torch.rand(1, 1, 4, 4, 1), torch.eye(4).view(1, 1, 4, 4))
    with pytest.raises(ValueError, match="`live_frame` must have poses"):
        slam._localize(gs.Pointclouds(), r, None)
    with pytest.raises(TypeError, match="Expected prev_frame to be of type gradslam.RGBDImages or None"):
        slam._localize(gs.Pointclouds(), r, 3)


def test_fusionutils_error_contracts():
    pc = gs.Pointclouds([torch.rand(4, 3)])
    r = gs.RGBDImages(torch.rand(1, 2, 4, 4, 3), torch.rand(1, 2, 4, 4, 1), torch.eye(4).view(1, 1, 4, 4),
                      torch.eye(4).view(1, 1, 4, 4).repeat(1, 2, 1, 1))
    tab = torch.zeros(2, 4, dtype=torch.int64)
    with pytest.raises(TypeError, match="Expected pointclouds to be of type gradslam.Pointclouds"):
        fusionutils.find_active_map_points(1, r)
    with pytest.raises(ValueError, match="Expected rgbdimages to have sequence length of 1"):
        fusionutils.find_active_map_points(pc, r)
    assert fusionutils.find_active_map_points(gs.Pointclouds(), r[:, 0]).shape == (0, 4)
    with pytest.raises(TypeError, match="Expected input pc2im_bnhw to have dtype of"):
        fusionutils.find_similar_map_points(pc, r[:, 0], tab.int(), 0.1, 0.5)
    with pytest.raises(ValueError, match="Expected pc2im_bnhw.ndim of 2"):
        fusionutils.find_similar_map_points(pc, r[:, 0], tab[0], 0.1, 0.5)
    with pytest.raises(ValueError, match="Pointclouds must have normals"):
        fusionutils.find_similar_map_points(pc, r[:, 0], tab, 0.1, 0.5)
    with pytest.raises(ValueError, match="Pointclouds must have features"):
        fusionutils.find_best_unique_correspondences(pc, r[:, 0], tab)
    with pytest.raises(ValueError, match="Pointclouds must have normals for map fusion"):
        fusionutils.fuse_with_map(pc, r[:, 0], tab, 0.6)
    with pytest.raises(TypeError, match="Expected input sigma to be of type"):
        fusionutils.get_alpha(torch.rand(3, 3), "s")
    with pytest.raises(ValueError, match="dimension to be 3"):
        fusionutils.get_alpha(torch.rand(3, 4), 0.6)
    with pytest.raises(ValueError, match="tensor1 and tensor2 should have the same shape"):
        fusionutils.are_points_close(torch.rand(3, 3), torch.rand(2, 3), 0.1)
    with pytest.warns(RuntimeWarning, match="Max of dot product was"):
        fusionutils.are_normals_similar(torch.ones(2, 3) * 2, torch.ones(2, 3), 0.5)
    assert fusionutils.are_points_close(torch.zeros(2, 3), torch.zeros(2, 3) + 0.01, 0.05).all()


# ------------------------------------------------------------------ geometry helpers
def test_se3_and_rigid_algebra():
    from gradslam_amd.geometry import geometryutils as gu
    from gradslam_amd.geometry import se3utils

    T = se3utils.se3_exp(torch.tensor([0.1, -0.2, 0.3, 0.3, 0.2, -0.1]))
    assert torch.allclose(T[:3, :3] @ T[:3, :3].t(), torch.eye(3), atol=1e-6) and T[3].tolist() == [0, 0, 0, 1]
    Ti = gu.inverse_transformation(T)
    assert torch.allclose(gu.compose_transformations(T, Ti), torch.eye(4), atol=1e-6)
    assert torch.allclose(gu.relative_transformation(T, T), torch.eye(4), atol=1e-5)
    small = se3utils.se3_exp(torch.tensor([1.0, 2.0, 3.0, 1e-8, 0.0, 0.0]))
    assert torch.allclose(small[:3, 3], torch.tensor([1.0, 2.0, 3.0]), atol=1e-6)
    g = gu.create_meshgrid(3, 4, normalized_coords=False)
    assert g.shape == (1, 3, 4, 2) and g[0, 2, 3].tolist() == [2.0, 3.0]


def test_pointclouds_plotly_export():
    """Viewer export (reference structures/pointclouds.py:1296-1383): host-side, works on any device."""
    import gradslam_amd as gs

    torch.manual_seed(0)
    pts, cols = torch.rand(2, 50, 3), torch.rand(2, 50, 3)
    pc = gs.Pointclouds(points=pts, colors=cols)
    sc = pc.plotly(1, as_figure=False)
    assert len(sc.x) == 50 and abs(float(sc.z[7]) - float(pts[1, 7, 2])) < 1e-7
    assert sc.marker.color.shape == (50, 3) and sc.marker.color.dtype.name == "uint8" and int(sc.marker.color.max()) > 200
    fig = pc.plotly(0, max_num_points=10)
    assert len(fig.data) == 1 and len(fig.data[0].x) == 10
    with pytest.raises(TypeError):
        pc.plotly("0")


def test_rgbdimages_plotly_export():
    """Animated viewer export (reference structures/rgbdimages.py:764-900, structutils.py:127-178)."""
    import gradslam_amd as gs

    torch.manual_seed(0)
    r = gs.RGBDImages(torch.rand(2, 3, 12, 16, 3), torch.rand(2, 3, 12, 16, 1) + 0.5, torch.eye(4).view(1, 1, 4, 4).repeat(2, 1, 1, 1))
    frames = r.plotly(1, as_figure=False)
    assert len(frames) == 3 and len(frames[0]["data"]) == 2 and frames[2]["name"] == 2
    assert frames[0]["data"][0].source.startswith("data:image/jpeg;base64,") and "depth" in frames[0]["data"][1].hovertemplate
    fig = r.plotly(0, include_depth=False)
    assert len(fig.frames) == 3 and len(fig.layout.sliders[0].steps) == 3
    with pytest.raises(TypeError):
        r.plotly(0.0)


def test_recorded_bench_line_follows_the_contract():
    """The bench line committed under profiles/ (what `python bench.py` printed on the MI355X) carries every key of
    the driver's contract, BASELINE.json's metric, and a consistent roofline / cpu_baseline."""
    import glob
    import json

    lines = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_bench.json")))
    assert lines
    d = json.load(open(lines[-1]))
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"].replace("x", "×") in base["metric"].replace("x", "×") or "frames/sec" in d["metric"]
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-2
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-2
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]


def test_require_hip_rejects_tensors_of_another_device(monkeypatch):
    """The C ABI takes raw pointers and the CURRENT device's stream: a tensor on another device must raise before any
    launch.  No GPU here: tensors are stand-ins with the two attributes the guard reads."""
    class _T:
        def __init__(self, index):
            self.is_cuda, self.device = True, torch.device("cuda", index)

    monkeypatch.setattr(_native, "_raw_device", lambda: 0)
    _native.require_hip(_T(0), None, _T(0), op="probe")
    with pytest.raises(RuntimeError, match=r"current HIP device is cuda:0.*set_device\(1\)"):
        _native.require_hip(_T(0), _T(1), op="probe")
    monkeypatch.setattr(_native, "_raw_device", lambda: 1)
    _native.require_hip(_T(1), op="probe")
    with pytest.raises(RuntimeError, match="tensor is on cuda:0"):
        _native.require_hip(_T(0), op="probe")


def test_sequence_tape_projection():
    """PointFusion.sequence_tape_bytes: the memory law of the one-node differentiable sequence (44 B per pixel and frame for
    the fusion tape, ~20 B per ds-grid point and association for the localisation tape, 4 B per target slot): the figure
    measured at configs[2] (~7.7 GB for 200 frames of 640x480, gradicp) and its scaling."""
    import gradslam_amd as gs

    pf = gs.slam.PointFusion(odom="gradicp", dsratio=4, numiters=10)
    full = pf.sequence_tape_bytes(200, 480, 640)
    assert 6e9 < full < 10e9
    assert pf.sequence_tape_bytes(100, 480, 640) < 0.6 * full
    gt = gs.slam.PointFusion(odom="gt")
    assert gt.sequence_tape_bytes(200, 480, 640) < 0.5 * full  # no localisation tape at all
    assert gs.slam.PointFusion(odom="icp", dsratio=4, numiters=10).sequence_tape_bytes(200, 480, 640) < full
