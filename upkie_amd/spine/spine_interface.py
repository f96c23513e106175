"""Agent end of the shared memory: same wire protocol and public methods as
the reference's ``SpineInterface``
(upkie/envs/backends/spine/spine_interface.py:20-199), so that either client
talks to either spine."""

import sys
import time
from multiprocessing import resource_tracker
from multiprocessing.shared_memory import SharedMemory
from time import perf_counter_ns

import msgpack

from ..exceptions import SpineError, UpkieTimeoutError
from .request import Request


def _serialize(obj):
    """msgpack `default=` hook, serialize.py:11-37: numpy arrays and friends become lists."""
    if hasattr(obj, "tolist"):
        return obj.tolist()
    if hasattr(obj, "np"):
        return obj.np.tolist()
    if hasattr(obj, "serialize"):
        return obj.serialize()
    return obj


def wait_for_shared_memory(shm_name: str, retries: int) -> SharedMemory:
    """wait_for_shared_memory.py:16-47: one attempt per second."""
    shm_name = shm_name.lstrip("/")
    for trial in range(retries):
        if trial > 0:
            time.sleep(1.0)
        try:
            shared_memory = SharedMemory(shm_name, size=0, create=False)
            try:  # the spine owns (and unlinks) the file
                resource_tracker.unregister(shared_memory._name, "shared_memory")
            except Exception:  # noqa: BLE001
                pass
            return shared_memory
        except FileNotFoundError:
            pass
    raise SpineError(f"spine /{shm_name} did not respond after {retries} attempts")


class SpineInterface:
    def __init__(self, shm_name: str = "/upkie", retries: int = 1, timeout_ns: int = 100_000_000):
        self._shared_memory = wait_for_shared_memory(shm_name, retries)
        self._buf = self._shared_memory.buf
        self._packer = msgpack.Packer(default=_serialize, use_bin_type=True)
        self._timeout_ns = timeout_ns

    def close(self) -> None:
        if getattr(self, "_shared_memory", None) is not None:
            self._buf = None
            self._shared_memory.close()  # the spine unlinks, spine_interface.py:57-66
            self._shared_memory = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # requests, spine_interface.py:68-106
    def set_action(self, action: dict) -> dict:
        self._wait_for_spine()
        self._write_dict(action)
        self._write_request(Request.kAction)
        self._wait_for_spine()
        return self._read_dict()

    def start(self, config: dict) -> dict:
        self._wait_for_spine()
        self._write_dict(config)
        self._write_request(Request.kStart)
        self._wait_for_spine()
        return self._read_dict()

    def stop(self) -> None:
        self._wait_for_spine()
        self._write_request(Request.kStop)

    # wire format, spine_interface.py:108-199
    def _read_request(self) -> int:
        return int.from_bytes(self._buf[0:4], sys.byteorder)

    def _write_request(self, request: int) -> None:
        self._buf[0:4] = int(request).to_bytes(4, sys.byteorder)

    def _read_dict(self) -> dict:
        assert self._read_request() == Request.kNone
        size = int.from_bytes(self._buf[4:8], sys.byteorder)
        return msgpack.unpackb(bytes(self._buf[8 : 8 + size]), raw=False)

    def _write_dict(self, dictionary: dict) -> None:
        assert self._read_request() == Request.kNone
        data = self._packer.pack(dictionary)
        self._buf[4:8] = len(data).to_bytes(4, sys.byteorder)
        self._buf[8 : 8 + len(data)] = data

    def _wait_for_spine(self) -> None:
        deadline = perf_counter_ns() + self._timeout_ns
        while self._read_request() not in (Request.kNone, Request.kError):
            if perf_counter_ns() > deadline:
                raise UpkieTimeoutError(
                    f"Spine did not process request within {self._timeout_ns / 1e6:.1f} ms, is it stopped?"
                )
            time.sleep(0)  # let a spine running in a thread of this process take the interpreter
        if self._read_request() == Request.kError:
            self._write_request(Request.kNone)
            raise SpineError("Invalid request, is the spine started?")
