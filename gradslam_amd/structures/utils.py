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
    per_b = [ops.select_rows_multi([vmap[b].reshape(-1, 3), nmap[b].reshape(-1, 3), rgb[b].reshape(-1, 3)],
                                   mask[b].reshape(-1)) for b in range(B)]
    return Pointclouds(points=[p[0] for p in per_b], normals=[p[1] for p in per_b], colors=[p[2] for p in per_b])
