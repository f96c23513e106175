"""Clamping helpers with the reference's semantics (upkie/utils/clamp.py)."""

import logging
from typing import Optional

logger = logging.getLogger("upkie_amd")


def clamp(value: float, lower: Optional[float] = None, upper: Optional[float] = None) -> float:
    """Clamp between optional bounds (clamp.py:15-29)."""
    if lower is not None and value < lower:
        return lower
    if upper is not None and value > upper:
        return upper
    return value


def clamp_abs(value: float, bound: float) -> float:
    """Clamp the absolute value, keeping the sign (clamp.py:32-39)."""
    return clamp(value, -bound, bound)


def clamp_and_warn(value: float, lower: float, upper: float, label: str) -> float:
    """Clamp and log when the value changes (clamp.py:42-58). NaN compares
    false with both bounds and passes through unchanged."""
    if value < lower:
        logger.warning("%s=%s clamped to lower=%s", label, value, lower)
        return lower
    if value > upper:
        logger.warning("%s=%s clamped to upper=%s", label, value, upper)
        return upper
    return value


def clamp_abs_and_warn(value: float, bound: float, label: str) -> float:
    """clamp.py:61-70."""
    return clamp_and_warn(value, -bound, bound, label)
