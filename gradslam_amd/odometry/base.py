"""Odometry provider interface (reference odometry/base.py:6-19) -- the plugin seam of the path."""
from abc import ABC, abstractmethod

__all__ = ["OdometryProvider"]


class OdometryProvider(ABC):
    """Subclass and override `provide()`; ICPSLAM calls
    ``provide(maps_pointclouds, frames_pointclouds) -> (B, 1, 4, 4)``."""

    def __init__(self, *params):
        pass

    @abstractmethod
    def provide(self, *args, **kwargs):
        raise NotImplementedError
