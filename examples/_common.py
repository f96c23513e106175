"""Shared by the examples: step counts can be shortened for smoke tests."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def steps(default: int) -> int:
    """Number of steps to run: EXAMPLE_STEPS overrides the example's own."""
    return int(os.environ.get("EXAMPLE_STEPS", default))
