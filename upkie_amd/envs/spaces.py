"""Observation/action spaces. gymnasium's classes are used when the package is
installed; otherwise minimal stand-ins with the same attributes (low, high,
shape, dtype, contains, sample) keep the API usable offline."""

import numpy as np

try:  # pragma: no cover - exercised only where gymnasium exists
    from gymnasium.spaces import Box, Dict  # type: ignore

    HAVE_GYMNASIUM = True
except ImportError:
    HAVE_GYMNASIUM = False

    class Box:  # type: ignore
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self, rng=None):
            rng = rng if rng is not None else np.random.default_rng()
            low = np.where(np.isfinite(self.low), self.low, -1.0)
            high = np.where(np.isfinite(self.high), self.high, 1.0)
            return rng.uniform(low, high).astype(self.dtype)

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"

    class Dict(dict):  # type: ignore
        def __init__(self, spaces=None):
            super().__init__(spaces or {})
            self.spaces = self

        def contains(self, x) -> bool:
            return all(k in x and self[k].contains(x[k]) for k in self)

        def sample(self, rng=None):
            return {k: s.sample(rng) for k, s in self.items()}


def batch_box(space, num_envs: int):
    """Batched version of a Box space (gymnasium.vector convention)."""
    return Box(
        np.repeat(space.low[None], num_envs, axis=0),
        np.repeat(space.high[None], num_envs, axis=0),
        shape=(num_envs,) + tuple(space.shape),
        dtype=space.dtype,
    )
