from .pointclouds import Pointclouds  # noqa: F401
from .rgbdimages import RGBDImages  # noqa: F401
from .utils import pointclouds_from_rgbdimages  # noqa: F401
