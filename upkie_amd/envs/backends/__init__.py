from .backend import Backend
from .hip_backend import HipBackend

__all__ = ["Backend", "HipBackend"]
