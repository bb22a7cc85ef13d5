"""pointclouds_from_rgbdimages (reference structures/utils.py:7-57)."""
from .. import ops
from .pointclouds import Pointclouds
from .rgbdimages import RGBDImages

__all__ = ["pointclouds_from_rgbdimages"]


def pointclouds_from_rgbdimages(rgbdimages: RGBDImages, *, global_coordinates: bool = True,
                                filter_missing_depths: bool = True) -> Pointclouds:
    """One-frame RGBDImages -> Pointclouds of its (valid) pixels in row-major order."""
    if not isinstance(rgbdimages, RGBDImages):
        raise TypeError("Expected rgbdimages to be of type gradslam.RGBDImages. Got {0}.".format(type(rgbdimages)))
    if not rgbdimages.shape[1] == 1:
        raise ValueError("Expected rgbdimages to have sequence length of 1. Got {0}.".format(rgbdimages.shape[1]))
    B = rgbdimages.shape[0]
    rgbdimages = rgbdimages.to_channels_last()
    vmap = rgbdimages.global_vertex_map if global_coordinates else rgbdimages.vertex_map
    nmap = rgbdimages.global_normal_map if global_coordinates else rgbdimages.normal_map
    rgb = rgbdimages.rgb_image
    if not filter_missing_depths:
        return Pointclouds(points=vmap.reshape(B, -1, 3).contiguous(), normals=nmap.reshape(B, -1, 3).contiguous(),
                           colors=rgb.reshape(B, -1, 3).contiguous())
    mask = rgbdimages.valid_depth_mask.squeeze(-1)  # (B,1,H,W)
    sel = lambda x, b: ops.mask_select(x[b].reshape(-1, 3), mask[b].reshape(-1))
    return Pointclouds(points=[sel(vmap, b) for b in range(B)], normals=[sel(nmap, b) for b in range(B)],
                       colors=[sel(rgb, b) for b in range(B)])
