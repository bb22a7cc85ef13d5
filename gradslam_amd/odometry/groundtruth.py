"""GroundTruthOdometryProvider (reference odometry/groundtruth.py:11-75): 4x4 algebra only."""
import torch

from ..geometry.geometryutils import relative_transformation
from ..structures.rgbdimages import RGBDImages
from .base import OdometryProvider

__all__ = ["GroundTruthOdometryProvider"]


class GroundTruthOdometryProvider(OdometryProvider):
    def __init__(self):
        pass

    def provide(self, rgbdimages1: RGBDImages, rgbdimages2: RGBDImages) -> torch.Tensor:
        """Relative transform between the two frames' ground-truth poses -> (B, 1, 4, 4)."""
        if not isinstance(rgbdimages1, RGBDImages):
            raise TypeError("Expected rgbdimages1 to be of type gradslam.RGBDImages. Got {0}.".format(type(rgbdimages1)))
        if not isinstance(rgbdimages2, RGBDImages):
            raise TypeError("Expected rgbdimages2 to be of type gradslam.RGBDImages. Got {0}.".format(type(rgbdimages2)))
        if not rgbdimages1.shape[1] == 1:
            raise ValueError("Sequence length of rgbdimages1 must be 1, but was {0}.".format(rgbdimages1.shape[1]))
        if not rgbdimages2.shape[1] == 1:
            raise ValueError("Sequence length of rgbdimages2 must be 1, but was {0}.".format(rgbdimages2.shape[1]))
        if rgbdimages1.shape[0] != rgbdimages2.shape[0]:
            raise ValueError("Batch size of rgbdimages1 and rgbdimages2 must be equal ({0} != {1}).".format(
                rgbdimages1.shape[0], rgbdimages2.shape[0]))
        return relative_transformation(rgbdimages1.poses.squeeze(1), rgbdimages2.poses.squeeze(1),
                                       orthogonal_rotations=False).unsqueeze(1)
