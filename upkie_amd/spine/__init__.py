"""Spine-side IPC: the shared-memory protocol between
an agent and a spine (upkie/cpp/spine/{AgentInterface,StateMachine,Spine}.cpp,
upkie/envs/backends/spine/spine_interface.py), with the GPU simulation playing
the role of the Bullet spine for ONE env of a batch. An unmodified
`Upkie-Spine-*` agent of the reference connects to `HipSpine` exactly as it
connects to `bullet_spine`."""

from .agent_interface import AgentInterface
from .hip_spine import HipSpine
from .request import Request
from .spine_interface import SpineInterface
from .state_machine import Event, State, StateMachine

__all__ = ["AgentInterface", "Event", "HipSpine", "Request", "SpineInterface", "State", "StateMachine"]
