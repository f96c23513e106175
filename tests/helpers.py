"""Shared helpers: matched (oracle, HIP) pairs built from one config."""

import numpy as np

from upkie_amd import abi
from upkie_amd.model.default_model import default_model


def randomized_config(num_envs: int, seed: int = 0, autoreset: bool = False):
    """Config of BASELINE.json configs[1]: pitch
    +-0.1 rad, x +-0.05 m, omega_y +-0.1 rad/s, v_x +-0.05 m/s."""
    cfg = abi.default_sim_config(num_envs, seed=seed)
    cfg.rand_pitch = 0.1
    cfg.rand_x = 0.05
    cfg.rand_omega_y = 0.1
    cfg.rand_linvel[0] = 0.05
    cfg.autoreset_mode = (
        abi.AUTORESET_NEXT_STEP if autoreset else abi.AUTORESET_DISABLED
    )
    return cfg


def make_pair(num_envs: int, seed: int = 0, autoreset: bool = False, cfg=None, model=None):
    """An oracle and a HIP simulation sharing one config and model."""
    from oracle import oracle as O
    from upkie_amd.sim import BatchedSim

    cfg = cfg if cfg is not None else randomized_config(num_envs, seed, autoreset)
    model = model if model is not None else default_model()
    return O.Oracle(model, cfg), BatchedSim(cfg, model)


def state_errors(oracle_state: np.ndarray, hip_state: np.ndarray) -> dict:
    """Max absolute error per state field between fp64 oracle and fp32 HIP."""
    fields = {
        "pos": (abi.S_POS, 3),
        "quat": (abi.S_QUAT, 4),
        "linvel": (abi.S_LINVEL, 3),
        "angvel": (abi.S_ANGVEL, 3),
        "q": (abi.S_Q, 6),
        "qd": (abi.S_QD, 6),
        "legref": (abi.S_LEGREF, 4),
        "yaw": (abi.S_YAW, 2),
        "torque": (abi.S_TORQUE, 6),
        "episode": (abi.S_EPISODE, 1),
        "done": (abi.S_DONE, 1),
        "contact": (abi.S_CONTACT, 1),
    }
    out = {}
    for name, (start, n) in fields.items():
        a = oracle_state[start : start + n]
        b = hip_state[start : start + n].astype(np.float64)
        out[name] = float(np.max(np.abs(a - b)))
    return out
