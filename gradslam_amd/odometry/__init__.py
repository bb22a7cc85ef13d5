from .base import OdometryProvider  # noqa: F401
from .gradicp import GradICPOdometryProvider  # noqa: F401
from .groundtruth import GroundTruthOdometryProvider  # noqa: F401
from .icp import ICPOdometryProvider  # noqa: F401
from . import icputils  # noqa: F401
