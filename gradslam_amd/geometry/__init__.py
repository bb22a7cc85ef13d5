from . import geometryutils, projutils, se3utils  # noqa: F401
from .projutils import *  # noqa: F401,F403  (reference geometry/__init__.py:1)
