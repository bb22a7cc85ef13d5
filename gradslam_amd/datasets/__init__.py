"""Dataset front-end of the hot path: TUM RGB-D sequences on disk -> tensors.

Same class / function names, arguments and outputs as the reference's gradslam.datasets for what is here
(datautils, tumutils, TUM, ICL, Scannet).  Frames are decoded on the host (PIL) and -- MI355X-first -- converted on the
device from their raw integer form (`TUM.load_rgbdimages`, gs_frames_from_raw): 5 bytes per pixel cross PCIe
instead of 16."""
from . import datautils, tumutils  # noqa: F401
from .icl import ICL  # noqa: F401
from .scannet import Scannet  # noqa: F401
from .tum import TUM  # noqa: F401
