"""Dataset front-end: datautils / tumutils / TUM on a synthetic TUM-layout tree written
by the test itself (no dataset ships with the image), known answers of the reference's own
tests/datasets/test_datautils.py restated as data, and -- on the GPU -- the device-side raw-frame conversion
against the host path."""
import os
import warnings

import numpy as np
import pytest
import torch

from gradslam_amd.datasets import datautils, tumutils
from gradslam_amd.datasets.tum import TUM

N_FRAMES, H, W = 9, 48, 64


def _quat(axis, ang):
    axis = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    return np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]])


def _rot(axis, ang):
    a = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


@pytest.fixture(scope="module")
def tum_tree(tmp_path_factory):
    from PIL import Image

    root = tmp_path_factory.mktemp("TUM")
    seq = root / "rgbd_dataset_freiburg1_synthetic"
    (seq / "rgb").mkdir(parents=True)
    (seq / "depth").mkdir()
    rng = np.random.default_rng(0)
    rgb_lines, depth_lines, gt_lines = ["# color images", "# timestamp filename"], ["# depth maps"], ["# ground truth", "# t tx ty tz qx qy qz qw"]
    truth = {"rgb": [], "depth": [], "pose": []}
    for i in range(N_FRAMES):
        t_rgb = 1305031102.175304 + i * 0.033
        t_depth = t_rgb + 0.004 + 0.001 * (i % 3)   # unsynchronised, within the 0.02 s radius
        t_pose = t_rgb - 0.002
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        depth = rng.integers(0, 30000, (H, W), dtype=np.uint16)
        depth[rng.random((H, W)) < 0.1] = 0
        Image.fromarray(rgb, "RGB").save(seq / "rgb" / ("%.6f.png" % t_rgb))
        Image.fromarray(depth).save(seq / "depth" / ("%.6f.png" % t_depth))
        rgb_lines.append("%.6f rgb/%.6f.png" % (t_rgb, t_rgb))
        depth_lines.append("%.6f depth/%.6f.png" % (t_depth, t_depth))
        axis, ang, trans = [0.2, 1.0, -0.3], 0.05 * i + 0.1, np.array([0.01 * i, -0.02 * i, 0.5 + 0.03 * i])
        q = _quat(axis, ang)
        gt_lines.append("%.4f %.6f %.6f %.6f %.8f %.8f %.8f %.8f" % (t_pose, *trans, *q))
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = _rot(axis, ang), trans
        truth["rgb"].append(rgb); truth["depth"].append(depth); truth["pose"].append(T)
    # a depth frame with no colour partner and a pose line that must be skipped
    depth_lines.append("%.6f depth/orphan.png" % (1305031102.175304 + 100.0))
    gt_lines.append("1305031300.0000 0 0 0 0 0 0 0")
    (seq / "rgb.txt").write_text("\n".join(rgb_lines) + "\n")
    (seq / "depth.txt").write_text("\n".join(depth_lines) + "\n")
    (seq / "groundtruth.txt").write_text("\n".join(gt_lines) + "\n")
    (seq / "accelerometer.txt").write_text("# unused\n")
    return str(root), truth


# ---------------------------------------------------------------------- datautils (reference test_datautils.py)
def test_normalize_and_channels_first():
    img = np.random.default_rng(1).integers(0, 256, (4, 6, 24, 32, 3), dtype=np.uint8)
    out = datautils.normalize_image(img)
    assert out.dtype == np.float64 and out.max() <= 1.0 and out.min() >= 0.0
    out = datautils.normalize_image(torch.from_numpy(img))
    assert out.dtype == torch.float32 and float(out.max()) <= 1.0
    with pytest.raises(TypeError):
        datautils.normalize_image([0, 125, 255])
    for x in (img, torch.from_numpy(img)):
        cf = datautils.channels_first(x)
        assert tuple(cf.shape) == (4, 6, 3, 24, 32) and cf.dtype == x.dtype
        assert np.array_equal(np.asarray(cf)[1, 2, 1], img[1, 2, :, :, 1])
    with pytest.raises(TypeError):
        datautils.channels_first([0, 125, 255])
    with pytest.raises(ValueError):
        datautils.channels_first(np.zeros((5, 10), dtype=np.uint8))
    with pytest.warns(UserWarning):
        assert datautils.channels_first(np.zeros((2, 10, 3), dtype=np.uint8)).shape == (3, 2, 10)


def test_scale_intrinsics_known_answers():
    syn = np.array([[10, 0, 5, 0], [0, 4, 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    gt = np.array([[2, 0, 1, 0], [0, 2, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    for conv in (lambda a: a, torch.tensor):
        out = datautils.scale_intrinsics(conv(syn), w_ratio=0.2, h_ratio=0.5)
        assert abs(np.asarray(out) - gt).sum() < 0.1
        back = datautils.scale_intrinsics(out, w_ratio=5.0, h_ratio=2.0)
        assert abs(np.asarray(back) - syn).sum() < 0.1
    K = np.array([[577.87, 0.0, 319.5, 0.0], [0.0, 577.87, 239.5, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    both = np.stack([K, K * np.array([[0.65, 1, 0.69, 1]] * 4)])
    np.testing.assert_allclose(datautils.scale_intrinsics(both, 2, 2)[0], datautils.scale_intrinsics(K, 2, 2))
    np.testing.assert_allclose(datautils.scale_intrinsics(both[:, :3, :3], 2, 2)[0], datautils.scale_intrinsics(K[:3, :3], 2, 2))
    with pytest.raises(TypeError):
        datautils.scale_intrinsics("abc", 2, 2)
    with pytest.raises(ValueError):
        datautils.scale_intrinsics(torch.rand(5, 10, 4, 3), 2, 2)
    with pytest.warns(UserWarning):
        datautils.scale_intrinsics(torch.rand(5, 10, 4, 4), 2, 2)


def test_pointquaternion_and_transforms():
    rng = np.random.default_rng(2)
    pq = rng.normal(size=(5, 3, 7))
    T = datautils.pointquaternion_to_homogeneous(pq)
    Tt = datautils.pointquaternion_to_homogeneous(torch.from_numpy(pq))
    assert T.shape == (5, 3, 4, 4) and T.dtype == np.float32 and Tt.dtype == torch.float32
    np.testing.assert_allclose(T, Tt.numpy(), atol=1e-6)
    for pqi, Ti in zip(pq.reshape(-1, 7), T.reshape(-1, 4, 4)):
        x, y, z, w = pqi[3:] / np.linalg.norm(pqi[3:])     # textbook unit-quaternion rotation
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(Ti[:3, :3], R, atol=2e-6)
        np.testing.assert_allclose(Ti[:3, 3], pqi[:3], rtol=1e-6)
        np.testing.assert_allclose(Ti[3], [0, 0, 0, 1])
        np.testing.assert_allclose(tumutils.transform44((0.0, *pqi)), Ti, atol=2e-6)
    with pytest.raises(TypeError):
        datautils.pointquaternion_to_homogeneous([1.0] * 7)
    with pytest.raises(TypeError):
        datautils.pointquaternion_to_homogeneous(pq, eps=1)
    with pytest.raises(ValueError):
        datautils.pointquaternion_to_homogeneous(pq[..., :6])
    tr = datautils.poses_to_transforms(list(T.reshape(-1, 4, 4)[:4].astype(np.float64)))
    np.testing.assert_allclose(tr[0], np.eye(4))
    np.testing.assert_allclose(T.reshape(-1, 4, 4)[1] @ tr[2], T.reshape(-1, 4, 4)[2], atol=1e-5)
    lab = datautils.create_label_image(np.array([[0, 1], [1, 2]]), {"a": (1, 2, 3), "b": (4, 5, 6), "c": (7, 8, 9)}.values())
    assert lab.dtype == np.uint8 and lab[1, 1].tolist() == [7, 8, 9] and lab[0, 1].tolist() == [4, 5, 6]


def test_associate_equals_all_pairs_search():
    """The bisection search must return what the reference's all-pairs formulation returns."""
    rng = np.random.default_rng(3)
    for trial in range(20):
        a = {"%.6f" % t: [str(t)] for t in np.sort(rng.uniform(0, 10, 60))}
        b = {"%.6f" % t: [str(t)] for t in np.sort(rng.uniform(0, 10, 70))}
        offset, md = float(rng.uniform(-0.05, 0.05)), float(rng.uniform(0.02, 0.3))
        cand = sorted((abs(float(x) - (float(y) + offset)), x, y) for x in a for y in b if abs(float(x) - (float(y) + offset)) < md)
        fa, fb, ref = set(a), set(b), []
        for _, x, y in cand:
            if x in fa and y in fb:
                fa.remove(x); fb.remove(y); ref.append((x, y))
        assert tumutils.associate(a, b, offset, md) == sorted(ref)


# ---------------------------------------------------------------------- TUM on the synthetic tree
def test_tum_sequences_and_values(tum_tree):
    root, truth = tum_tree
    ds = TUM(root, seqlen=3, dilation=1, stride=2, height=H, width=W)
    # frames 0,2,4 | 2,4,6 | 4,6,8
    assert len(ds) == 3
    color, depth, K, poses, transforms, names, stamps = ds[1]
    assert color.shape == (3, H, W, 3) and depth.shape == (3, H, W, 1) and K.shape == (1, 4, 4)
    assert color.dtype == depth.dtype == poses.dtype == transforms.dtype == torch.float32
    for j, f in enumerate((2, 4, 6)):
        assert torch.equal(color[j], torch.from_numpy(truth["rgb"][f]).float())
        assert torch.equal(depth[j, ..., 0], torch.from_numpy((truth["depth"][f].astype(np.int64) / 5000.0)).float())
    np.testing.assert_allclose(K[0].numpy(), [[525.0 * W / 640, 0, 319.5 * W / 640, 0], [0, 525.0 * H / 480, 239.5 * H / 480, 0],
                                              [0, 0, 1, 0], [0, 0, 0, 1]], rtol=1e-6)
    P = [truth["pose"][f] for f in (2, 4, 6)]
    np.testing.assert_allclose(poses[0].numpy(), np.eye(4), atol=1e-5)
    np.testing.assert_allclose(poses[2].numpy(), np.linalg.inv(P[0]) @ P[2], atol=2e-5)
    np.testing.assert_allclose(transforms[0].numpy(), np.eye(4), atol=1e-6)
    np.testing.assert_allclose(transforms[2].numpy(), np.linalg.inv(P[1]) @ P[2], atol=2e-5)
    assert names.count(",") == 2 and names.startswith("rgbd_dataset_freiburg1_synthetic/")
    assert stamps.count("\n") == 2 and stamps.startswith("rgb 1305031102.2")
    # options
    ds2 = TUM(root, sequences=("rgbd_dataset_freiburg1_synthetic",), seqlen=2, start=1, end=7, channels_first=True,
              normalize_color=True, return_pose=False, return_transform=False, return_names=False, return_timestamps=False)
    assert len(ds2) == 3
    color, depth, K = ds2[0]
    assert color.shape == (2, 3, 480, 640) and depth.shape == (2, 1, 480, 640) and float(color.max()) <= 1.0
    assert torch.equal(depth[0, 0, ::10, ::10], torch.from_numpy(truth["depth"][1][::1, ::1].astype(np.int64) / 5000.0).float()[
        np.minimum(np.floor(np.arange(0, 480, 10) * (H / 480)).astype(int), H - 1)][:, np.minimum(np.floor(np.arange(0, 640, 10) * (W / 640)).astype(int), W - 1)])


def test_tum_errors(tum_tree, tmp_path):
    root, _ = tum_tree
    with pytest.raises(TypeError):
        TUM(root, seqlen=2.0)
    with pytest.raises(TypeError):
        TUM(root, sequences=["rgbd_dataset_freiburg1_synthetic"])
    with pytest.raises(ValueError):
        TUM(root, sequences=())
    with pytest.raises(ValueError):
        TUM(root, sequences=("rgbd_dataset_freiburg2_missing",))
    with pytest.raises(ValueError):
        TUM(root, start=5, end=3)
    (tmp_path / "not_a_tum_folder").mkdir()
    with pytest.raises(ValueError):
        TUM(str(tmp_path))


# ---------------------------------------------------------------------- device path
@pytest.mark.gpu
def test_device_frame_conversion_equals_host_path(tum_tree):
    import gradslam_amd as gs

    gs._native.lib()
    root, _ = tum_tree
    ds = TUM(root, seqlen=4, height=H, width=W)
    color, depth, K, poses = ds[0][:4]
    frames = ds.load_rgbdimages(0, "cuda:0")
    assert frames.shape == (1, 4, H, W)
    assert torch.equal(frames.rgb_image[0].cpu(), color) and torch.equal(frames.depth_image[0].cpu(), depth)
    assert torch.equal(frames.intrinsics[0].cpu(), K) and torch.allclose(frames.poses[0].cpu(), poses, atol=1e-6)
    # resized + normalised + channels first: same arithmetic on both sides (bilinear colour, nearest depth)
    ds = TUM(root, seqlen=2, height=96, width=80, normalize_color=True, channels_first=True)
    color, depth = ds[1][:2]
    frames = ds.load_rgbdimages(1, "cuda:0")
    assert frames.channels_first and frames.rgb_image.shape == (1, 2, 3, 96, 80)
    assert torch.equal(frames.depth_image[0].cpu(), depth)
    torch.testing.assert_close(frames.rgb_image[0].cpu(), color, rtol=0, atol=2e-7)
    # and the loaded sequence runs through the hot path
    slam = gs.slam.PointFusion(odom="gt", device="cuda:0")
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pcs, rec = slam(TUM(root, seqlen=3, height=H, width=W).load_rgbdimages(0, "cuda:0"))
    assert rec.shape == (1, 3, 4, 4) and int(pcs.num_points_per_pointcloud.item()) > 0


def test_helpers_against_reference_golden(golden, tmp_path):
    """tests/golden/ref_datasets.npz = outputs of the reference's own datautils / tumutils (tools/gen_golden_datasets.py)."""
    g = golden("ref_datasets")
    np.testing.assert_array_equal(datautils.pointquaternion_to_homogeneous(g["pq"].copy()), g["pq_T"])
    np.testing.assert_array_equal(datautils.scale_intrinsics(g["K"], 0.25, 0.5), g["K_scaled"])
    np.testing.assert_array_equal(np.stack(datautils.poses_to_transforms(list(g["poses"]))), g["poses_transforms"])
    fa, fb, ft = (str(tmp_path / n) for n in ("rgb.txt", "depth.txt", "groundtruth.txt"))
    open(fa, "w").write("# colour\n" + "\n".join("%.6f rgb/%.6f.png" % (t, t) for t in g["stamps_a"]) + "\n")
    open(fb, "w").write("# depth\n" + "\n".join("%.6f depth/%.6f.png" % (t, t) for t in g["stamps_b"]) + "\n")
    open(ft, "w").write("# gt\n" + "\n".join("%.4f %.6f %.6f %.6f %.6f %.6f %.6f %.6f" % tuple(r) for r in g["traj"]) + "\n")
    da, db = tumutils.read_file_list(fa, 3, 150), tumutils.read_file_list(fb)
    for tag, off, md in (("m0", 0.0, 0.02), ("m1", 0.013, 0.05), ("m2", -0.02, 0.3)):
        m = tumutils.associate(da, db, off, md)
        np.testing.assert_array_equal([float(x) for x, _ in m], g[tag + "_a"])
        np.testing.assert_array_equal([float(y) for _, y in m], g[tag + "_b"])
    tr = tumutils.read_trajectory(ft, matrix=True)
    np.testing.assert_array_equal([float(k) for k in tr.keys()], g["traj_keys"])
    np.testing.assert_array_equal(np.stack(list(tr.values())), g["traj_T"])


# ---------------------------------------------------------------------- ICL on a synthetic tree
@pytest.fixture(scope="module")
def icl_tree(tmp_path_factory):
    from PIL import Image

    root = tmp_path_factory.mktemp("ICL")
    rng = np.random.default_rng(5)
    truth = {}
    for traj, n in ((0, 7), (2, 6)):
        tdir = root / ("living_room_traj%d_frei_png" % traj)
        (tdir / "rgb").mkdir(parents=True)
        (tdir / "depth").mkdir()
        assoc, sim = [], []
        truth[traj] = {"rgb": [], "depth": [], "pose": []}
        for i in range(n):
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            depth = rng.integers(1, 40000, (H, W), dtype=np.uint16)
            Image.fromarray(rgb, "RGB").save(tdir / "rgb" / ("%d.png" % i))
            Image.fromarray(depth).save(tdir / "depth" / ("%d.png" % i))
            assoc.append("%d depth/%d.png %d rgb/%d.png" % (i, i, i, i))
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = _rot([0.3, -1.0, 0.2], 0.03 * i + 0.2), [0.1 * i, 0.02 * i, -0.05 * i]
            truth[traj]["rgb"].append(rgb); truth[traj]["depth"].append(depth); truth[traj]["pose"].append(T)
            if not (traj == 0 and i == n - 1):  # traj0's pose file is one pose short, like the real one
                sim += [" ".join("%.8f" % v for v in T[r]) for r in range(3)] + [""]
        (tdir / "associations.txt").write_text("\n".join(assoc) + "\n")
        (tdir / ("livingRoom%dn.gt.sim" % traj)).write_text("\n".join(sim) + "\n")
    return str(root), truth


def test_icl_sequences_and_values(icl_tree):
    from gradslam_amd.datasets import ICL

    root, truth = icl_tree
    ds = ICL(root, seqlen=3, stride=3, height=H, width=W)
    # traj0: 7 frames minus the dropped last one = 6 -> 2 sequences; traj2: 6 -> 2 sequences
    assert len(ds) == 4
    color, depth, K, poses, transforms, names = ds[3]      # traj2, frames 3,4,5
    assert color.shape == (3, H, W, 3) and depth.shape == (3, H, W, 1) and names.count(",") == 2
    for j, f in enumerate((3, 4, 5)):
        assert torch.equal(color[j], torch.from_numpy(truth[2]["rgb"][f]).float())
        assert torch.equal(depth[j, ..., 0], torch.from_numpy(truth[2]["depth"][f].astype(np.int64) / 5000.0).float())
    np.testing.assert_allclose(K[0].numpy(), [[481.2 * W / 640, 0, 319.5 * W / 640, 0], [0, -480.0 * H / 480, 239.5 * H / 480, 0],
                                              [0, 0, 1, 0], [0, 0, 0, 1]], rtol=1e-6)
    P = [truth[2]["pose"][f] for f in (3, 4, 5)]
    np.testing.assert_allclose(poses[0].numpy(), np.eye(4), atol=1e-5)
    np.testing.assert_allclose(poses[2].numpy(), np.linalg.inv(P[0]) @ P[2], atol=2e-5)
    np.testing.assert_allclose(transforms[1].numpy(), np.linalg.inv(P[0]) @ P[1], atol=2e-5)
    assert len(ICL(root, trajectories=("living_room_traj2_frei_png",), seqlen=2, start=1, end=5, return_pose=False,
                   return_transform=False)) == 2
    with pytest.raises(ValueError):
        ICL(root, trajectories=("living_room_traj7_frei_png",))
    with pytest.raises(ValueError):
        ICL(root, trajectories=("not_a_trajectory",))
    with pytest.raises(TypeError):
        ICL(root, trajectories=["living_room_traj2_frei_png"])


# ---------------------------------------------------------------------- ScanNet on a synthetic tree
def test_scannet_sequences_labels_and_tables(tmp_path):
    from PIL import Image

    from gradslam_amd.datasets import Scannet
    from gradslam_amd.datasets.scannet import get_color_encoding, nyu40_to_scannet20

    rng = np.random.default_rng(9)
    base, meta = tmp_path / "scans", tmp_path / "sequence_associations"
    meta.mkdir()
    truth = {}
    for scene in ("scene0000_00", "scene0001_00"):
        for sub in ("color", "depth", "pose", "label-filt", "intrinsic"):
            (base / scene / sub).mkdir(parents=True)
        Kd = np.array([[577.6, 0, 318.9, 0], [0, 578.7, 242.7, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
        np.savetxt(base / scene / "intrinsic" / "intrinsic_depth.txt", Kd)
        lines, truth[scene] = [], {"rgb": [], "depth": [], "pose": [], "label": []}
        for i in range(5):
            rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            depth = rng.integers(0, 9000, (H, W), dtype=np.uint16)
            label = rng.integers(0, 41, (H, W), dtype=np.uint8)
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = _rot([0.1, 0.9, 0.4], 0.04 * i + 0.3), [0.2 * i, -0.1 * i, 0.05 * i]
            Image.fromarray(rgb, "RGB").save(base / scene / "color" / ("%d.png" % i))   # (real ScanNet: .jpg)
            Image.fromarray(depth).save(base / scene / "depth" / ("%d.png" % i))
            Image.fromarray(label).save(base / scene / "label-filt" / ("%d.png" % i))
            np.savetxt(base / scene / "pose" / ("%d.txt" % i), T)
            f = lambda sub, ext: "%s/%s/%d.%s" % (scene, sub, i, ext)
            lines.append("color %s depth %s pose %s label-filt %s a b c d e f intrinsic_depth %s/intrinsic/intrinsic_depth.txt"
                         % (f("color", "png"), f("depth", "png"), f("pose", "txt"), f("label-filt", "png"), scene))
            for k, v in (("rgb", rgb), ("depth", depth), ("pose", T), ("label", label)):
                truth[scene][k].append(v)
        (meta / ("%s-seq_0.txt" % scene)).write_text("\n".join(lines) + "\n")
    ds = Scannet(str(base), str(meta), scenes=("scene0001_00",), start=1, end=4, height=H, width=W)
    assert len(ds) == 1
    color, depth, K, poses, transforms, name, labels = ds[0]
    t = truth["scene0001_00"]
    assert name == "scene0001_00-seq_0" and color.shape == (3, H, W, 3) and labels.shape == (3, H, W, 1)
    for j, f in enumerate((1, 2, 3)):
        assert torch.equal(color[j], torch.from_numpy(t["rgb"][f]).float())
        assert torch.equal(depth[j, ..., 0], torch.from_numpy(t["depth"][f].astype(np.int64) / 1000.0).float())
        assert torch.equal(labels[j, ..., 0], torch.from_numpy(nyu40_to_scannet20(t["label"][f].copy())).float())
    np.testing.assert_allclose(K[0].numpy(), [[577.6 * W / 640, 0, 318.9 * W / 640, 0], [0, 578.7 * H / 480, 242.7 * H / 480, 0],
                                              [0, 0, 1, 0], [0, 0, 0, 1]], rtol=1e-6)
    np.testing.assert_allclose(poses[2].numpy(), np.linalg.inv(t["pose"][1]) @ t["pose"][3], atol=2e-5)
    np.testing.assert_allclose(transforms[2].numpy(), np.linalg.inv(t["pose"][2]) @ t["pose"][3], atol=2e-5)
    assert len(Scannet(str(base), str(meta), None, return_labels=False, seg_classes="nyu40")) == 2
    # class tables: 41 NYU40 classes, 21 benchmark classes (incl. unlabeled), shared colours
    nyu, s20 = get_color_encoding("nyu40"), get_color_encoding("scannet20")
    assert len(nyu) == 41 and len(s20) == 21 and all(nyu[k] == v for k, v in s20.items())
    assert list(s20)[:4] == ["unlabeled", "wall", "floor", "cabinet"] and list(s20)[-1] == "otherfurniture"
    lab = np.arange(41, dtype=np.uint8)
    out = nyu40_to_scannet20(lab.copy())
    assert out.max() == 20 and out[39] == 20 and out[13] == 0 and out[14] == 13 and out[12] == 12
    with pytest.raises(ValueError):
        Scannet(str(base), str(meta), None, start=3, end=2)
    with pytest.raises(ValueError):
        Scannet(str(base), str(meta), None, start=0, end=9)
