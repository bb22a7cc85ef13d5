"""gradslam_amd -- MI355X-native hot path of gradslam's dense SLAM: per-frame point-to-plane ICP and
the PointFusion map update, as hand-written HIP kernels (gfx950) behind gradslam's own Python API.

Drop-in surface (same names, arguments, warnings and errors as the reference):

    from gradslam_amd import RGBDImages, Pointclouds
    from gradslam_amd.slam import PointFusion, ICPSLAM
    from gradslam_amd.odometry import ICPOdometryProvider, GradICPOdometryProvider

The hot path runs only on HIP devices and only through gradslam_amd/libgradslam_hip.so; there is no
CPU or PyTorch fallback (importing works anywhere; computing raises without the library / a GPU).
"""
__version__ = "0.1.0"

from . import geometry, odometry, slam, structures  # noqa: F401,E402
from .geometry.projutils import *  # noqa: F401,F403,E402  (reference __init__.py:6)
from .structures import Pointclouds, RGBDImages  # noqa: F401,E402
