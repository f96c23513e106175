"""Bookkeeping of external forces for the envs and the backend:
``PyBulletBackend.set_external_forces`` semantics (pybullet_backend.py:603-658)
on top of `BatchedSim.set_external_forces`."""

from typing import Dict

import torch

from ..abi import MAX_EXTERNAL_FORCES
from ..exceptions import UpkieRuntimeError


class ExternalForceSet:
    """Forces per link name. As in the reference, an entry persists (and is
    re-applied at every substep) until the same link is given a new force; a
    zero force is how a push ends."""

    def __init__(self, model, num_envs: int):
        self._model = model
        self._num_envs = int(num_envs)
        self._entries: Dict[str, tuple] = {}

    def update(self, external_forces: Dict[str, object]) -> None:
        """`external_forces`: {link name: ExternalForce | (force, local) | force},
        force = 3-vector (every env) or ``[B, 3]`` tensor (one per env)."""
        staged = {}
        for link_name, spec in external_forces.items():
            if link_name not in self._model.link_names:  # pybullet_backend.py:613-617
                raise UpkieRuntimeError(f"Robot does not have a link named '{link_name}'")
            if hasattr(spec, "force"):
                force, local = spec.force, bool(spec.local)
            elif isinstance(spec, tuple):
                force, local = spec[0], bool(spec[1])
            else:
                force, local = spec, False
            force = torch.as_tensor(force, dtype=torch.float32)
            if force.shape == (3,):
                force = force.expand(self._num_envs, 3)
            if tuple(force.shape) != (self._num_envs, 3):  # external_force.py:38-41
                raise ValueError(f"Force must be a 3D vector, got shape {tuple(force.shape)}")
            staged[link_name] = (force, local)
        merged = dict(self._entries)
        merged.update(staged)
        if len(merged) > MAX_EXTERNAL_FORCES:
            raise UpkieRuntimeError(f"at most {MAX_EXTERNAL_FORCES} links can carry an external force at a time")
        self._entries = merged

    def clear(self) -> None:
        self._entries = {}

    def push(self, sim) -> None:
        """Hand the current set to the simulation."""
        if not self._entries:
            sim.set_external_forces(None)
            return
        forces, bodies, points, local = [], [], [], []
        for link_name, (force, is_local) in self._entries.items():
            body, point, rotation = self._model.link_attachment(link_name)
            if is_local:  # link frame -> frame of the composite body the link is welded into
                force = force @ torch.as_tensor(rotation, dtype=torch.float32).t()
            forces.append(force.t())
            bodies.append(body)
            points.append(point)
            local.append(is_local)
        sim.set_external_forces(torch.stack(forces).contiguous(), bodies=bodies, points=points, local=local)
