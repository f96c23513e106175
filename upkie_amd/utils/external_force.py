"""External force data structure (upkie/utils/external_force.py:11-55)."""

from typing import List, Union

import numpy as np


class ExternalForce:
    """Force applied to a robot link at its frame origin."""

    def __init__(self, force: Union[List[float], np.ndarray], local: bool = False):
        force = np.array(force, dtype=np.float64)
        if force.shape != (3,):  # external_force.py:38-41
            raise ValueError(f"Force must be a 3D vector, got shape {force.shape}")
        self.force = force
        self.local = local

    def __repr__(self) -> str:
        return f"ExternalForce(force={self.force.tolist()}, local={self.local})"
