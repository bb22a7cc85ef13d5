"""The reference's own scenario tests for the path, replayed against the HIP build.

Each test names the reference test it replays (file:line under the reference's tests/).  Inputs are the
reference fixture (tests/golden/msrd_b2s3.npz = its tests/data/msrd_b2s3 arrays) or the small hand-made
vectors those tests hold; the assertions are the reference's.  The CUDA-only reference tests
(test_icputils.py:284-387, :537-640) are included -- here they run on the MI355X.
The projection helpers are plain tensor algebra (not the hot path) and are checked on the CPU too."""
import numpy as np
import pytest
import torch

from tests.helpers import t

DEV = "cuda:0"


@pytest.fixture(scope="module")
def gs():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import gradslam_amd

    gradslam_amd._native.lib()  # fail loudly if the extension is missing
    return gradslam_amd


def _frames(gsmod, g, nb=2, channels_first=False):
    c, dp, K, P = (t(g[k][:nb]).to(DEV) for k in ("colors", "depths", "intrinsics", "poses"))
    if channels_first:
        c, dp = c.permute(0, 1, 4, 2, 3).contiguous(), dp.permute(0, 1, 4, 2, 3).contiguous()
    return gsmod.RGBDImages(c, dp, K, P, channels_first=channels_first)


def _first_frame_map(gsmod, frame, sigma):
    """Map made of one frame, confidence = alpha of the camera-frame points (test_fusionutils.py:15-24)."""
    from gradslam_amd.slam import fusionutils
    from gradslam_amd.structures.utils import pointclouds_from_rgbdimages

    glob = pointclouds_from_rgbdimages(frame)
    loc = pointclouds_from_rgbdimages(frame, global_coordinates=False)
    alpha = fusionutils.get_alpha(loc.points_padded, sigma)
    glob.features_padded = (alpha * glob.nonpad_mask.to(alpha.dtype)).unsqueeze(-1)
    return glob


HAND_A = [[5.0, 5.0, 5.0], [3.0, 3.0, 3.0], [1.0, 2.0, 3.0], [3.0, 2.0, 1.0], [-1.0, 0.0, 1.0], [0.0, 0.0, 0.0]]
HAND_B = [[1.0, 3.0, 5.0], [3.0, 2.0, 2.0], [1.0, 2.0, 3.0], [1.0, 2.0, 1.0], [1.0, 0.0, -1.0], [0.0, 0.0, 0.0]]


# ------------------------------------------------------------------ slam/test_fusionutils.py
@pytest.mark.gpu
def test_are_points_close(gs):
    """test_fusionutils.py:122-156: the threshold is strict, sqrt(2) apart is not close at dist_th sqrt(2)."""
    from gradslam_amd.slam import fusionutils

    a, b = torch.tensor(HAND_A, device=DEV), torch.tensor(HAND_B, device=DEV)
    assert fusionutils.are_points_close(a, b, 2.0 ** 0.5).int().tolist() == [0, 0, 1, 0, 0, 1]
    assert fusionutils.are_points_close(a, b, 2.01 ** 0.5).int().tolist() == [0, 1, 1, 0, 0, 1]


@pytest.mark.gpu
def test_are_normals_similar_and_warning(gs):
    """test_fusionutils.py:207-241 and :296-303 (unnormalised input warns)."""
    from gradslam_amd.slam import fusionutils

    a, b = torch.tensor(HAND_A[:5], device=DEV), torch.tensor(HAND_B[:5], device=DEV)
    a = a / a.pow(2).sum(-1, keepdim=True).sqrt()
    b = b / b.pow(2).sum(-1, keepdim=True).sqrt()
    assert fusionutils.are_normals_similar(a, b, 0.879).int().tolist() == [0, 1, 1, 0, 0]
    assert fusionutils.are_normals_similar(a, b, 0.878).int().tolist() == [1, 1, 1, 0, 0]
    with pytest.warns(RuntimeWarning, match="Max of dot product was "):
        fusionutils.are_normals_similar(torch.tensor([[3.0, 3.0, 3.0], [1.0, 2.0, 3.0]], device=DEV),
                                        torch.tensor([[3.0, 2.0, 2.0], [1.0, 2.0, 3.0]], device=DEV), 0.879)


@pytest.mark.gpu
def test_active_points_reproject_onto_their_own_pixels(gs, golden):
    """test_fusionutils.py:307-333: a map made of frame 0, looked up from frame 0, hits every valid pixel once
    and carries that pixel's colour."""
    from gradslam_amd.slam import fusionutils

    g = golden("msrd_b2s3")
    frames = _frames(gs, g)
    f0 = frames[:, 0]
    pc = _first_frame_map(gs, f0, 0.6)
    table = fusionutils.find_active_map_points(pc, f0)
    assert table.shape[0] == int(frames.valid_depth_mask[:, 0].sum())
    colors = t(g["colors"]).to(DEV)
    painted = torch.zeros_like(colors)
    painted[table[:, 0], 0, table[:, 2], table[:, 3]] = pc.colors_padded[table[:, 0], table[:, 1]]
    torch.testing.assert_close(painted[:, 0:1], colors[:, 0:1] * frames.valid_depth_mask[:, 0:1].float())


@pytest.mark.gpu
def test_similar_points_drop_only_zero_normals(gs, golden):
    """test_fusionutils.py:441-480: against its own frame only the valid-depth / zero-normal points fail."""
    from gradslam_amd.slam import fusionutils

    frames = _frames(gs, golden("msrd_b2s3"))
    f0 = frames[:, 0]
    pc = _first_frame_map(gs, f0, 0.6)
    active = fusionutils.find_active_map_points(pc, f0)
    similar, is_similar = fusionutils.find_similar_map_points(pc, f0, active, 0.05 ** 0.5, 0.9)
    dropped = active[~is_similar]
    nm = frames.normal_map
    assert float(nm[dropped[:, 0], 0, dropped[:, 2], dropped[:, 3]].abs().sum()) == 0  # possibly no rows at all
    zero_normals = sum(int(n.eq(0).all(-1).sum()) for n in pc.normals_list)
    assert active.shape[0] - similar.shape[0] == zero_normals
    assert all(int(p.eq(0).all(-1).sum()) == 0 for p in pc.points_list)


@pytest.mark.gpu
def test_correspondence_count(gs, golden):
    """test_fusionutils.py:881-914."""
    from gradslam_amd.slam import fusionutils

    frames = _frames(gs, golden("msrd_b2s3"))
    f0 = frames[:, 0]
    pc = _first_frame_map(gs, f0, 0.6)
    table = fusionutils.find_correspondences(pc, f0, 0.05 ** 0.5, 0.9)
    invalid = (~frames.valid_depth_mask[:, 0]).squeeze(-1).int()
    valid_zero_normals = frames.normal_map[:, 0].eq(0).all(-1).int() - invalid
    assert int(valid_zero_normals.abs().sum()) == int(valid_zero_normals.sum())
    assert int((frames.vertex_map[:, 0].eq(0).all(-1).int() - invalid).abs().sum()) == 0
    assert table.shape[0] == int(frames.valid_depth_mask[:, 0].sum()) - int(valid_zero_normals.sum())


@pytest.mark.gpu
def test_fuse_with_nothing_to_append(gs):
    """test_fusionutils.py:989-1040: an all-zero depth frame merges its matches and appends no rows."""
    from gradslam_amd.slam import fusionutils

    pts = torch.tensor(HAND_A, device=DEV).unsqueeze(0)
    table = torch.tensor([[0, 1, 0, 0], [0, 2, 0, 1], [0, 4, 1, 1], [0, 5, 1, 0]], device=DEV, dtype=torch.int64)
    image = torch.tensor([[[0.0, 1.0, 0.0], [0.0, 2.0, 0.0]], [[0.0, 5.0, 1.0], [8.0, 8.0, 8.0]]], device=DEV)[None, None]
    depths = torch.zeros_like(image[..., 0:1])
    K = torch.rand(4, 4)[None, None].to(DEV)
    P = torch.eye(4)[None, None].to(DEV)
    frame = gs.RGBDImages(image, depths, K, P, channels_first=False)
    pc = gs.Pointclouds(points=pts, normals=pts, colors=pts, features=torch.ones_like(pts[..., 0:1]))
    out = fusionutils.fuse_with_map(pc, frame, table, 0.6)
    assert int(out.num_points_per_pointcloud[0]) == 6


@pytest.mark.gpu
def test_update_map_fusion_grows_and_looser_thresholds_fuse_more(gs, golden):
    """test_fusionutils.py:1140-1177."""
    from gradslam_amd.slam import fusionutils

    frames = _frames(gs, golden("msrd_b2s3"))
    counts = []
    for dist_th, dot_th in ((0.05 ** 0.5, 0.9), (0.4 ** 0.5, 0.5)):
        pc = _first_frame_map(gs, frames[:, 0], 0.6)
        before = pc.num_points_per_pointcloud.clone()
        pc = fusionutils.update_map_fusion(pc, frames[:, 1], dist_th, dot_th, 0.6)
        after = pc.num_points_per_pointcloud
        assert after.gt(before).all()
        counts.append(after.clone())
    assert counts[0].gt(counts[1]).all()


# ------------------------------------------------------------------ odometry/test_icputils.py (CUDA-only there)
def _known_transform_case(gsmod, g, axis, rad):
    from gradslam_amd.structures.utils import pointclouds_from_rgbdimages

    src = pointclouds_from_rgbdimages(_frames(gsmod, g, nb=1)[:, 0])
    c, s = float(np.cos(rad)), float(np.sin(rad))
    R = {"x": [[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]], "z": [[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]}[axis]
    T = torch.eye(4, device=DEV)
    T[:3, :3] = torch.tensor(R, device=DEV)
    T[:3, 3] = torch.tensor([0.05, 0.03, 0.01], device=DEV)
    return src, src.transform(T), T


@pytest.mark.gpu
@pytest.mark.parametrize("axis", ["x", "z"])
@pytest.mark.parametrize("fn", ["point_to_plane_ICP", "point_to_plane_gradICP"])
def test_icp_recovers_transform_100_iterations(gs, golden, fn, axis):
    """test_icputils.py:286-387 (ICP transform1/2) and :539-640 (gradICP transform1/2): 0.2 rad about x / z,
    100 iterations, no distance threshold, default torch.testing tolerances of the reference's assert_allclose
    for fp32 (rtol 1e-4, atol 1e-5 as SURVEY 8c quotes them)."""
    from gradslam_amd.odometry import icputils

    src, tgt, T = _known_transform_case(gs, golden("msrd_b2s3"), axis, 0.2)
    out, idx = getattr(icputils, fn)(src.points_padded, tgt.points_padded, tgt.normals_padded,
                                     torch.eye(4, device=DEV), 100, 1e-8, None)
    assert out.shape == T.shape
    torch.testing.assert_close(out, T, rtol=1e-4, atol=1e-5)
    # the clouds are the same points moved rigidly: at convergence every source point's neighbour is itself
    n = src.points_padded.shape[1]
    assert (idx.reshape(-1)[:n].cpu() == torch.arange(n)).float().mean() > 0.999


# ------------------------------------------------------------------ odometry/test_groundtruth.py
@pytest.mark.gpu
def test_groundtruth_provider(gs):
    """test_groundtruth.py:52-83."""
    from gradslam_amd.odometry.groundtruth import GroundTruthOdometryProvider

    def rot(rad):
        return [[np.cos(rad), -np.sin(rad), 0.0, 0.05], [np.sin(rad), np.cos(rad), 0.0, 0.03],
                [0.0, 0.0, 1.0, 0.01], [0.0, 0.0, 0.0, 1.0]]

    poses = torch.tensor([rot(0.1), rot(0.7)], dtype=torch.float32).unsqueeze(0)
    frames = gs.RGBDImages(torch.rand(1, 2, 32, 32, 3).to(DEV), torch.rand(1, 2, 32, 32, 1).to(DEV),
                           torch.rand(1, 1, 4, 4).to(DEV), poses.to(DEV))
    odom = GroundTruthOdometryProvider()
    rel = odom.provide(frames[:, 0], frames[:, 1])
    assert rel.shape == frames[:, 1].poses.shape
    torch.testing.assert_close(frames[:, 0].poses.squeeze() @ rel.squeeze(), frames[:, 1].poses.squeeze())
    with pytest.raises(TypeError):
        odom.provide(frames[:, 0], torch.rand(1, 1, 4, 4))
    with pytest.raises(TypeError):
        odom.provide(torch.rand(1, 1, 4, 4), frames[:, 0])


# ------------------------------------------------------------------ structures/test_utils.py
@pytest.mark.gpu
@pytest.mark.parametrize("channels_first", [False, True])
def test_pointclouds_from_rgbdimages_project_back_to_the_pixel_grid(gs, golden, channels_first):
    """structures/test_utils.py:18-66: the cloud of a frame projects back onto the centres of its valid pixels,
    and the unfiltered cloud contains the filtered one in order."""
    from gradslam_amd.geometry.geometryutils import create_meshgrid
    from gradslam_amd.structures.utils import pointclouds_from_rgbdimages

    g = golden("msrd_b2s3")
    frames = _frames(gs, g, channels_first=channels_first)
    f0 = frames[:, 0]
    pc = pointclouds_from_rgbdimages(f0)
    K = t(g["intrinsics"]).to(DEV).squeeze(1)
    proj0 = pc.pinhole_projection(K).points_list[0][..., :-1]
    h, w = frames.shape[2], frames.shape[3]  # RGBDImages.shape is (B, L, H, W) in either layout
    grid = create_meshgrid(h, w, False).to(DEV).squeeze(0)
    grid = torch.cat([grid[..., 1:], grid[..., 0:1]], -1)
    mask0 = frames.valid_depth_mask[0, 0]
    mask0 = mask0.squeeze(0) if channels_first else mask0.squeeze(-1)
    torch.testing.assert_close(proj0.round().float(), grid[mask0].float())

    full = pointclouds_from_rgbdimages(f0, filter_missing_depths=False)
    for b in range(len(pc)):
        kept, every = pc.points_list[b], full.points_list[b]
        # the reference walks both lists; equivalently the filtered rows are the unfiltered rows at valid pixels
        m = frames.valid_depth_mask[b, 0].reshape(-1)
        assert every.shape[0] == m.numel()
        assert ((kept - every[m]) ** 2).sum(-1).max() < 1e-12


# ------------------------------------------------------------------ geometry/test_projutils.py (CPU algebra)
@pytest.mark.parametrize("lastdim", [3, 4])
def test_project_points_shapes_and_errors(lastdim):
    """test_projutils.py:95-194."""
    import gradslam_amd as gsm

    assert gsm.project_points(torch.rand(10, lastdim), torch.rand(4, 4)).shape == (10, 2)
    assert gsm.project_points(torch.rand(2, 10, lastdim), torch.rand(4, 4)).shape == (2, 10, 2)
    assert gsm.project_points(torch.rand(2, 10, lastdim), torch.rand(2, 4, 4)).shape == (2, 10, 2)
    with pytest.raises(TypeError):
        gsm.project_points([1, 2, 3], torch.rand(4, 4))
    with pytest.raises(TypeError):
        gsm.project_points(torch.rand(10, lastdim), [1, 2, 3])
    for bad_pts in (torch.rand(2), torch.rand(2, 2), torch.rand(2, 5)):
        with pytest.raises(ValueError):
            gsm.project_points(bad_pts, torch.rand(4, 4))
    for bad_mat in (torch.rand(4), torch.rand(4, 3), torch.rand(3, 4), torch.rand(3, 3)):
        with pytest.raises(ValueError):
            gsm.project_points(torch.rand(10, lastdim), bad_mat)
    with pytest.raises(ValueError):  # batched matrix, unbatched points
        gsm.project_points(torch.rand(10, lastdim), torch.rand(1, 4, 4))
    with pytest.raises(ValueError):  # batch sizes differ
        gsm.project_points(torch.rand(2, 10, lastdim), torch.rand(3, 4, 4))


def test_project_points_values():
    import gradslam_amd as gsm

    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 500.0, 400.0, 320.0, 240.0
    pts = torch.tensor([[0.1, -0.2, 2.0], [0.0, 0.0, 1.0], [1.0, 1.0, 0.0]])
    uv = gsm.project_points(pts, K)
    torch.testing.assert_close(uv[:2], torch.tensor([[345.0, 200.0], [320.0, 240.0]]))
    torch.testing.assert_close(uv[2], torch.tensor([500.0, 400.0]))  # z == 0: divide by one


@pytest.mark.parametrize("lastdim", [2, 3])
def test_unproject_points_shapes_values_and_errors(lastdim):
    """test_projutils.py:198-267."""
    import gradslam_amd as gsm

    assert gsm.unproject_points(torch.rand(10, lastdim), torch.rand(3, 3), torch.rand(10)).shape == (10, 3)
    assert gsm.unproject_points(torch.rand(2, 10, lastdim), torch.rand(3, 3), torch.rand(2, 10)).shape == (2, 10, 3)
    assert gsm.unproject_points(torch.rand(2, 10, lastdim), torch.rand(2, 3, 3), torch.rand(2, 10)).shape == (2, 10, 3)
    with pytest.raises(TypeError):
        gsm.unproject_points([1, 2, 3], [1, 2, 3], [1, 2, 3])
    with pytest.raises(TypeError):
        gsm.unproject_points(torch.rand(2, 10, lastdim), [1, 2, 3], [1, 2, 3])
    with pytest.raises(TypeError):
        gsm.unproject_points(torch.rand(2, 10, lastdim), torch.rand(2, 3, 3), [1, 2, 3])
    with pytest.raises(ValueError):
        gsm.unproject_points(torch.rand(2), torch.rand(3, 3), torch.rand(2, 10))
    with pytest.raises(ValueError):
        gsm.unproject_points(torch.rand(2, 3), torch.rand(3), torch.rand(2, 10))
    with pytest.raises(ValueError):
        gsm.unproject_points(torch.rand(2, 3), torch.rand(3, 3), torch.rand(1))
    with pytest.raises(ValueError):
        gsm.unproject_points(torch.rand(2, 1, 2, 3), torch.rand(1, 3, 3), torch.rand(2, 1, 2))
    # round trip through the pinhole model
    K = torch.tensor([[500.0, 0.0, 320.0], [0.0, 400.0, 240.0], [0.0, 0.0, 1.0]])
    K4 = torch.eye(4)
    K4[:3, :3] = K
    pts = torch.rand(7, 3) + torch.tensor([0.0, 0.0, 1.0])
    uv = gsm.project_points(pts, K4)
    back = gsm.unproject_points(uv, torch.inverse(K), pts[:, 2])
    torch.testing.assert_close(back, pts, rtol=1e-4, atol=1e-5)


def test_homogenize_round_trip_and_points_at_infinity():
    """test_projutils.py:12-91."""
    import gradslam_amd as gsm

    pts = torch.rand(5, 4, 3)
    h = gsm.homogenize_points(pts)
    assert h.shape == (5, 4, 4) and bool((h[..., -1] == 1).all())
    torch.testing.assert_close(gsm.unhomogenize_points(h), pts)
    inf = torch.tensor([[2.0, 4.0, 0.0], [2.0, 4.0, 2.0]])
    torch.testing.assert_close(gsm.unhomogenize_points(inf), torch.tensor([[2.0, 4.0], [1.0, 2.0]]))
    for fn in (gsm.homogenize_points, gsm.unhomogenize_points):
        with pytest.raises(TypeError):
            fn([1.0, 2.0])
        with pytest.raises(ValueError):
            fn(torch.rand(3))


@pytest.mark.parametrize("lastdim", [3, 4])
def test_inverse_intrinsics(lastdim):
    """test_projutils.py:270-345."""
    import gradslam_amd as gsm

    vals = torch.rand(5, 10, 4) + 0.1
    K = torch.zeros(5, 10, lastdim, lastdim)
    K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2] = vals.unbind(-1)
    K[..., 2, 2] = 1
    K[..., -1, -1] = 1
    inv = gsm.inverse_intrinsics(K)
    ref = torch.linalg.inv(K)
    assert inv.shape == K.shape
    assert float((inv - ref).abs().sum() / ref.abs().sum()) < 1e-2
    with pytest.raises(TypeError):
        gsm.inverse_intrinsics([1, 2, 3])
    for bad in (torch.rand(3), torch.rand(3, 4), torch.rand(5, 5)):
        with pytest.raises(ValueError):
            gsm.inverse_intrinsics(bad)


# ------------------------------------------------------------------ structures/test_pointclouds.py (container algebra, CPU)
def _ragged3():
    return [torch.tensor([[0.1, 0.3, 0.5], [0.5, 0.2, 0.1], [0.6, 0.8, 0.7]]),
            torch.tensor([[0.1, 0.3, 0.3], [0.6, 0.7, 0.8], [0.2, 0.3, 0.4], [0.1, 0.5, 0.3]]),
            torch.tensor([[0.7, 0.3, 0.6], [0.2, 0.4, 0.8], [0.9, 0.5, 0.2], [0.2, 0.3, 0.4], [0.9, 0.3, 0.8]])]


def _expect(pc, want_list):
    for got, want in zip(pc.points_list, want_list):
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    for b, want in enumerate(want_list):
        n = want.shape[0]
        torch.testing.assert_close(pc.points_padded[b, :n], want, rtol=1e-5, atol=1e-6)
        assert bool((pc.points_padded[b, n:] == 0).all())  # padding stays zero under every op


def test_pointclouds_arithmetic_operators():
    """structures/test_pointclouds.py:153-269: scalar, (1,1,3) and (B,1,3) operands; results own their storage."""
    import gradslam_amd as gsm

    pts = _ragged3()
    pc = gsm.Pointclouds([p.clone() for p in pts])
    pc.offset_(5)
    _expect(pc, [p + 5 for p in pts])
    pc = gsm.Pointclouds([p.clone() for p in pts])
    _expect(pc + 5, [p + 5 for p in pts])
    a = torch.tensor([2.0, 5.0, 7.0]).reshape(1, 1, 3)
    _expect(pc + a, [p + a.squeeze() for p in pts])
    per_b = torch.tensor([[0.0, 1.0, -4.0], [-2.0, 10.0, -0.2], [1.0, 5.0, 3.0]]).unsqueeze(1)
    res = pc + per_b
    _expect(res, [p + per_b[b] for b, p in enumerate(pts)])
    pc.scale_(5)
    _expect(pc, [p * 5 for p in pts])
    assert res.points_padded.data_ptr() != pc.points_padded.data_ptr()
    pc = gsm.Pointclouds([p.clone() for p in pts])
    _expect(pc * a, [p * a.squeeze() for p in pts])
    _expect(pc - a, [p - a.squeeze() for p in pts])
    _expect((pc * -1) + a, [a.squeeze() - p for p in pts])
    _expect(pc / a, [p / a.squeeze() for p in pts])
    _expect(pc, pts)  # the out-of-place operators left the operand alone


def test_pointclouds_rigid_and_projective_ops():
    """structures/test_pointclouds.py:271-455: rotate / transform (single and per-cloud matrices, pre and post
    multiplication, normals rotate with the points) and pinhole projection (single and per-cloud intrinsics)."""
    import gradslam_amd as gsm

    pts = _ragged3()
    T = torch.tensor([[-0.802837, 0.056561, -0.593509, 2.583219], [0.596192, 0.071654, -0.799638, 4.008804],
                      [-0.002701, -0.995825, -0.091248, 1.439254], [0.0, 0.0, 0.0, 1.0]])
    R, tv = T[:3, :3], T[:3, 3]
    new = lambda: gsm.Pointclouds([p.clone() for p in pts])  # noqa: E731
    pre = [p @ R.t() for p in pts]
    _expect(new().rotate_(R), pre)
    both = gsm.Pointclouds([p.clone() for p in pts], [p * 2 for p in pts]).rotate(R)
    _expect(both, pre)
    for got, p in zip(both.normals_list, pts):
        torch.testing.assert_close(got, (p * 2) @ R.t(), rtol=1e-5, atol=1e-6)
    post = [p @ R for p in pts]
    _expect(new().rotate_(R, pre_multiplication=False), post)
    _expect(new() @ R, post)
    Rb = torch.stack([R * i for i in (1, 2, 3)])
    pre_b = [p @ Rb[b].t() for b, p in enumerate(pts)]
    _expect(new().rotate_(Rb), pre_b)
    _expect(new().rotate(Rb), pre_b)
    _expect(new().transform_(T), [p + tv for p in pre])
    _expect(new().transform(T), [p + tv for p in pre])
    Tb = torch.stack([T * i for i in (1, 2, 3)])
    moved_b = [p + tv * (b + 1) for b, p in enumerate(pre_b)]
    _expect(new().transform_(Tb), moved_b)
    _expect(new().transform(Tb), moved_b)

    def K(f, cx, cy):
        return torch.tensor([[f, 0.0, cx, 0.0], [0.0, f, cy, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])

    Ks = [K(577.87, 319.5, 239.5), K(377.87, 219.5, 139.5), K(677.87, 419.5, 339.5)]
    singles = [new().pinhole_projection_(k) for k in Ks]
    for s, k in zip(singles, Ks):
        assert s.points_padded.shape == (3, 5, 3) and s.num_points_per_pointcloud.tolist() == [3, 4, 5]
        _expect(s, [torch.stack([k[0, 0] * p[:, 0] / p[:, 2] + k[0, 2], k[1, 1] * p[:, 1] / p[:, 2] + k[1, 2],
                                 torch.ones(len(p))], -1) for p in pts])
    batched = new().pinhole_projection_(torch.stack(Ks))
    for b in range(3):
        torch.testing.assert_close(batched.points_padded[b], singles[b].points_padded[b])
    for bad in (torch.eye(3)[None], torch.rand(2, 3, 3), "R"):
        with pytest.raises((TypeError, ValueError)):
            new().rotate_(bad)


def test_pointclouds_clone_detach_index_append_empty():
    """structures/test_pointclouds.py:729-1042, :1266-1300: clone / detach semantics, every index form, append
    onto ragged and empty containers."""
    import gradslam_amd as gsm

    pts = [p.clone().requires_grad_(True) for p in _ragged3()]
    pc = gsm.Pointclouds(pts, [p * 2 for p in pts], [p * 3 for p in pts], [p[:, :1] for p in pts])
    c = pc.clone()
    assert c.points_padded.requires_grad and c.points_padded.data_ptr() != pc.points_padded.data_ptr()
    torch.testing.assert_close(c.colors_padded, pc.colors_padded)
    dt = pc.detach()
    assert not dt.points_padded.requires_grad and not dt.features_list[0].requires_grad
    for index, want in ((1, [1]), (slice(0, 2), [0, 1]), ([2, 0], [2, 0]), (torch.tensor([0, 2]), [0, 2]),
                        (torch.tensor([False, True, True]), [1, 2])):
        sub = pc[index]
        assert len(sub) == len(want)
        for got, b in zip(sub.normals_list, want):
            torch.testing.assert_close(got, pts[b] * 2)
    with pytest.raises(IndexError):
        pc[3.5]
    grown = pc.detach().clone()
    grown.append_points(pc.detach())
    assert grown.num_points_per_pointcloud.tolist() == [6, 8, 10]
    for b in range(3):
        torch.testing.assert_close(grown.points_list[b], torch.cat([pts[b], pts[b]]).detach())
        torch.testing.assert_close(grown.features_list[b], torch.cat([pts[b][:, :1]] * 2).detach())
    empty = gsm.Pointclouds()
    assert len(empty) == 0 and not empty.has_points and empty.points_list is None and empty.points_padded is None
    empty.append_points(pc.detach())
    assert empty.num_points_per_pointcloud.tolist() == [3, 4, 5] and empty.has_normals and empty.has_colors
