"""Request word at offset 0 of the shared memory (upkie/cpp/spine/Request.h,
upkie/envs/backends/spine/request.py:10-44): set by the agent, reset to
`kNone` by the spine once processed."""

from enum import IntEnum


class Request(IntEnum):
    kNone = 0  # no active request
    kAction = 1  # an action has been supplied
    kStart = 2  # start the spine with the configuration supplied
    kStop = 3  # stop the spine
    kError = 4  # the last request was invalid
