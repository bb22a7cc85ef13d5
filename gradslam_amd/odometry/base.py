"""Odometry provider interface (reference odometry/base.py:6-19) -- the plugin seam of the path.

The contract ICPSLAM relies on (slam/icpslam.py:243):
``provide(maps_pointclouds: Pointclouds, frames_pointclouds: Pointclouds) -> torch.Tensor (B, 1, 4, 4)``,
the rigid transform that aligns every frame cloud to its map cloud."""
import abc

__all__ = ["OdometryProvider"]


class OdometryProvider(abc.ABC):
    def __init__(self, *params):
        """Providers keep their own parameters; the base class has no state."""

    @abc.abstractmethod
    def provide(self, *args, **kwargs):
        """One odometry estimate per batch element; concrete providers define the arguments."""
        raise NotImplementedError
