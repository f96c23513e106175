"""Exception types of the host layer.

Same class names and hierarchy as the reference's ``upkie/exceptions.py:7-54``
so that agents written against ``upkie.exceptions`` catch the same errors.
"""


class UpkieException(Exception):
    """Base class for exceptions raised by this package."""


class FallDetected(UpkieException):
    """Raised when a fall is detected."""


class MissingOptionalDependency(UpkieException):
    """An optional feature lacks its optional dependency."""


class ModelError(UpkieException):
    """Something is wrong in the robot model."""


class PerformanceIssue(UpkieException):
    """A performance issue was detected (upkie/exceptions.py:31-34)."""


class SpineError(UpkieException):
    """A spine did not answer or refused a request (upkie/exceptions.py:37-42)."""


class UpkieRuntimeError(UpkieException, RuntimeError):
    """Runtime error, for instance an invalid call to a library function."""


class UpkieTimeoutError(UpkieException, TimeoutError):
    """Raised when something times out."""
