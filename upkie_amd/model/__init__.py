"""Robot model (upkie/model/__init__.py:6-23): the names agents import.

`KinematicTree`, `Link`, `CollisionGeometry` and `SE3` of the reference are
what its `Model` is built from; here the URDF is reduced in one pass
(`urdf.py`) to the 7-body structure the kernels integrate, so only the classes
an agent or an env touches exist."""

from .joint_properties import JointProperties
from .model import Joint, JointLimit, Model

__all__ = ["Joint", "JointLimit", "JointProperties", "Model"]
