"""Filters with the reference's semantics (upkie/utils/filters.py)."""

from typing import Tuple

import numpy as np


def low_pass_filter(prev_output: float, cutoff_period: float, new_input: float, dt: float) -> float:
    """First-order low-pass (filters.py:63-80); asserts the Nyquist bound."""
    alpha = dt / cutoff_period
    assert alpha < 0.5  # Nyquist-Shannon sampling theorem
    return prev_output + alpha * (new_input - prev_output)


def abs_bounded_derivative_filter(
    prev_output: float, new_input: float, dt: float, max_output: float, max_derivative: float
) -> float:
    """Bounded output and derivative (filters.py:15-36)."""
    return bounded_derivative_filter(
        prev_output, new_input, dt, (-max_output, max_output), (-max_derivative, max_derivative)
    )


def bounded_derivative_filter(
    prev_output: float,
    new_input: float,
    dt: float,
    output_bounds: Tuple[float, float],
    derivative_bounds: Tuple[float, float],
) -> float:
    """filters.py:39-60."""
    derivative = (new_input - prev_output) / dt
    derivative = np.clip(derivative, *derivative_bounds)
    output = prev_output + derivative * dt
    return float(np.clip(output, *output_bounds))
