from . import geometryutils, projutils, se3utils  # noqa: F401
