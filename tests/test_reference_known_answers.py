"""The hand-made known-answer cases the reference's own tests hold for this path, restated as data
(inputs + expected outputs) and checked against BOTH the CPU oracle (always) and the HIP path (-m gpu).

Sources (reference tests/):
  odometry/test_icputils.py:18-49     solve_linear_system
  odometry/test_icputils.py:800-867   downsample_pointclouds
  odometry/test_icputils.py:942-1013  downsample_rgbdimages
  slam/test_fusionutils.py:27-53      get_alpha
  slam/test_fusionutils.py:672-750    find_best_unique_correspondences (sorting / tie-break)
  slam/test_fusionutils.py:918-986    fuse_with_map
  slam/test_fusionutils.py:988-1040   fuse_with_map with all-zero depth (nothing appended, no error)
"""
import pytest
import torch

from oracle import fusion as ofu
from oracle import icp as oicp
from oracle import maps as omaps
from oracle.cloud import Cloud

DEV = "cuda:0"

PTS6 = torch.tensor([[5.0, 5.0, 5.0], [3.0, 3.0, 3.0], [1.0, 2.0, 3.0], [3.0, 2.0, 1.0], [-1.0, 0.0, 1.0], [0.0, 0.0, 0.0]])
IMAGE22 = torch.tensor([[[0.0, 1.0, 0.0], [0.0, 2.0, 0.0]], [[0.0, 5.0, 1.0], [8.0, 8.0, 8.0]]]).view(1, 1, 2, 2, 3)

LIN_A = torch.tensor([[0.1, 0.7, 0.3, 0.6], [0.5, 0.2, 0.4, 0.8], [0.3, 0.9, 0.5, 0.2], [0.8, 0.2, 0.3, 0.4], [0.7, 0.9, 0.3, 0.8]])
LIN_B = torch.tensor([[0.7], [0.2], [0.9], [0.2], [0.9]])

ALPHA_EPS = 1e-20
ALPHA_GT = torch.tensor([ALPHA_EPS, 5.17e-17, 3.5924e-09, 3.5924e-09, 6.2177e-02, 1.0])

UNI_PTS = torch.tensor([[5.0, 5.0, 5.0], [3.0, 3.0, 3.0], [1.0, 2.0, 3.0], [-0.5, -0.5, 1.0], [-1.0, 0.0, 1.0], [0.0, 0.0, 0.0]])
UNI_TABLE = torch.tensor([[0, 4, 0, 0], [0, 0, 1, 1], [0, 5, 1, 0], [0, 1, 0, 0], [0, 2, 1, 1], [0, 3, 0, 0]])
UNI_K = torch.tensor([[2.0, 0.0, 1.0, 0.0], [0.0, 2.0, 1.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]).view(1, 1, 4, 4)
UNI_GT = torch.tensor([[0, 4, 0, 0], [0, 5, 1, 0], [0, 2, 1, 1]])

FUSE_TABLE = torch.tensor([[0, 1, 0, 0], [0, 2, 0, 1], [0, 5, 1, 0]])
FUSE_COLORS_GT = torch.tensor([[5.0, 5.0, 5.0], [1.5, 2.0, 1.5], [0.5, 2.0, 1.5], [3.0, 2.0, 1.0], [-1.0, 0.0, 1.0],
                               [0.0, 2.5, 0.5], [8.0, 8.0, 8.0]])
FUSE_TABLE_ZERO = torch.tensor([[0, 1, 0, 0], [0, 2, 0, 1], [0, 4, 1, 1], [0, 5, 1, 0]])

DS_PTS = torch.tensor([[5.0, 5.0, 5.0], [3.0, 3.0, 3.0], [1.0, 2.0, 3.0], [3.0, 2.0, 1.0], [1.0, 0.0, 1.0], [0.0, 0.0, 0.0]])
DS_TABLE = torch.tensor([[0, 0, 0, 0], [0, 1, 4, 2], [0, 2, 3, 1], [0, 3, 0, 3], [0, 4, 3, 3], [0, 5, 3, 6]])
DS3_GT = torch.tensor([[5.0, 5.0, 5.0], [3.0, 2.0, 1.0], [1.0, 0.0, 1.0], [0.0, 0.0, 0.0]])
DS2_GT = torch.tensor([[5.0, 5.0, 5.0], [3.0, 3.0, 3.0]])

DSI_IMAGE = torch.arange(12, dtype=torch.float32).view(3, 4, 1).repeat(1, 1, 3).view(1, 1, 3, 4, 3)
DSI_PTS_GT = torch.tensor([[0.0, 0.0, 1.0], [2.0, 0.0, 1.0], [0.0, 2.0, 1.0], [2.0, 2.0, 1.0]])
DSI_COL_GT = torch.tensor([[0.0] * 3, [2.0] * 3, [8.0] * 3, [10.0] * 3])


def close(a, b):
    torch.testing.assert_close(a.cpu(), b, rtol=1e-4, atol=1e-5)


# ====================================================================== oracle (CPU)
def test_oracle_solve_linear_system():
    x = oicp.solve_linear_system(LIN_A, LIN_B, 1e-8)
    # the reference asserts A x == b to float32 assert_allclose defaults on this (consistent) system
    torch.testing.assert_close(LIN_A @ x, LIN_B, rtol=1e-4, atol=1e-5)


def test_oracle_get_alpha():
    a = ofu.get_alpha(PTS6, 0.6, eps=ALPHA_EPS)
    close(a, ALPHA_GT)
    assert a.gt(0).all()


def _uni_frame_oracle():
    depth = torch.ones(1, 1, 2, 2, 1)
    V, N, gV, gN = omaps.all_maps(depth, UNI_K, None)
    return dict(rgb=IMAGE22, depth=depth, K=UNI_K, pose=None, V=V, N=N, gV=gV, gN=gN)


def test_oracle_unique_sorting():
    feats = ofu.get_alpha(UNI_PTS.unsqueeze(0), 0.6, keepdim=True)
    feats[0, 3] = 1e-12
    m = Cloud([UNI_PTS], None, None, [feats[0]])
    out = ofu.find_best_unique_correspondences(m, _uni_frame_oracle(), UNI_TABLE)
    assert torch.equal(out, UNI_GT)


def _fuse_frame_oracle(depth_value):
    depth = torch.ones(1, 1, 2, 2, 1) * depth_value
    torch.manual_seed(0)
    K = torch.rand(4, 4).view(1, 1, 4, 4)
    pose = torch.eye(4).view(1, 1, 4, 4)
    return ofu.make_frame(IMAGE22, depth, K, pose)


def test_oracle_fuse_with_map():
    m = Cloud([PTS6.clone()], [PTS6.clone()], [PTS6.clone()], [torch.ones(6, 1)])
    out = ofu.fuse_with_map(m, _fuse_frame_oracle(1e-20), FUSE_TABLE, 0.6)
    close(out.colors[0], FUSE_COLORS_GT)
    m = Cloud([PTS6.clone()], [PTS6.clone()], [PTS6.clone()], [torch.ones(6, 1)])
    out = ofu.fuse_with_map(m, _fuse_frame_oracle(0.0), FUSE_TABLE_ZERO, 0.6)
    assert out.counts == [6]  # zero depth: nothing appended, no error


def test_oracle_downsample():
    m = Cloud([DS_PTS], [DS_PTS * -1], [DS_PTS * 2])
    out = oicp.downsample_map(m, DS_TABLE, 3)
    close(out.points[0], DS3_GT), close(out.normals[0], DS3_GT * -1), close(out.colors[0], DS3_GT * 2)
    out = oicp.downsample_map(Cloud([DS_PTS]), DS_TABLE, 2)
    close(out.points[0], DS2_GT)
    depth = torch.ones(1, 1, 3, 4, 1)
    eye = torch.eye(4).view(1, 1, 4, 4)
    fr = ofu.make_frame(DSI_IMAGE, depth, eye, eye)
    out = oicp.downsample_frame(fr["gV"], fr["gN"], fr["rgb"], fr["depth"], 2)
    close(out.points[0], DSI_PTS_GT), close(out.colors[0], DSI_COL_GT)
    close(out.normals[0], fr["N"][0, 0, ::2, ::2].reshape(-1, 3))


# ====================================================================== HIP path (GPU)
@pytest.fixture(scope="module")
def gs():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import gradslam_amd

    gradslam_amd._native.lib()
    return gradslam_amd


@pytest.mark.gpu
def test_gpu_solve_linear_system(gs):
    x = gs.odometry.icputils.solve_linear_system(LIN_A.to(DEV), LIN_B.to(DEV), 1e-8)
    torch.testing.assert_close((LIN_A.to(DEV) @ x).cpu(), LIN_B, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_gpu_get_alpha(gs):
    a = gs.slam.fusionutils.get_alpha(PTS6.to(DEV), 0.6, eps=ALPHA_EPS)
    assert a.shape == ALPHA_GT.shape
    close(a, ALPHA_GT)
    assert a.gt(0).all()


@pytest.mark.gpu
def test_gpu_unique_sorting(gs):
    fu = gs.slam.fusionutils
    pts = UNI_PTS.unsqueeze(0).to(DEV)
    feats = fu.get_alpha(pts, 0.6, keepdim=True)
    feats[0, 3] = 1e-12
    pc = gs.Pointclouds(points=pts, features=feats)
    r = gs.RGBDImages(IMAGE22.to(DEV), torch.ones(1, 1, 2, 2, 1, device=DEV), UNI_K.to(DEV))
    close(r.vertex_map, torch.tensor([[[-0.5, -0.5, 1.0], [0.0, -0.5, 1.0]], [[-0.5, 0.0, 1.0], [0.0, 0.0, 1.0]]]).view(1, 1, 2, 2, 3))
    out = fu.find_best_unique_correspondences(pc, r, UNI_TABLE.to(DEV))
    assert torch.equal(out.cpu(), UNI_GT)


@pytest.mark.gpu
def test_gpu_fuse_with_map(gs):
    fu = gs.slam.fusionutils
    torch.manual_seed(0)
    K = torch.rand(4, 4).view(1, 1, 4, 4).to(DEV)
    pose = torch.eye(4).view(1, 1, 4, 4).to(DEV)
    pts = PTS6.unsqueeze(0).to(DEV)
    for depth_value, table, n_expected in ((1e-20, FUSE_TABLE, 7), (0.0, FUSE_TABLE_ZERO, 6)):
        r = gs.RGBDImages(IMAGE22.to(DEV), torch.ones(1, 1, 2, 2, 1, device=DEV) * depth_value, K, pose)
        pc = gs.Pointclouds(points=pts.clone(), normals=pts.clone(), colors=pts.clone(), features=torch.ones_like(pts[..., :1]))
        out = fu.fuse_with_map(pc, r, table.to(DEV), 0.6)
        assert out.colors_padded.shape == (1, n_expected, 3)
        if n_expected == 7:
            close(out.colors_padded[0], FUSE_COLORS_GT)


@pytest.mark.gpu
def test_gpu_downsample(gs):
    ut = gs.odometry.icputils
    pts = DS_PTS.unsqueeze(0).to(DEV)
    out = ut.downsample_pointclouds(gs.Pointclouds(pts, pts * -1, pts * 2), DS_TABLE.to(DEV), 3)
    assert out.points_padded.shape == (1, 4, 3)
    close(out.points_padded[0], DS3_GT), close(out.normals_padded[0], DS3_GT * -1), close(out.colors_padded[0], DS3_GT * 2)
    out = ut.downsample_pointclouds(gs.Pointclouds(pts), DS_TABLE.to(DEV), 2)
    assert out.normals_padded is None
    close(out.points_padded[0], DS2_GT)
    eye = torch.eye(4).view(1, 1, 4, 4).to(DEV)
    r = gs.RGBDImages(DSI_IMAGE.to(DEV), torch.ones(1, 1, 3, 4, 1, device=DEV), eye, eye)
    out = ut.downsample_rgbdimages(r, 2)
    close(out.points_padded[0], DSI_PTS_GT), close(out.colors_padded[0], DSI_COL_GT)
    close(out.normals_padded[0], r.normal_map[0, 0, ::2, ::2].reshape(-1, 3).cpu())
