"""Spine end of the shared memory (upkie/cpp/spine/AgentInterface.cpp:33-103):
creates ``/dev/shm/<name>`` exclusively, lays it out as
``[uint32 request][uint32 size][size bytes of MessagePack]`` and unlinks it when
closed."""

import sys
from multiprocessing.shared_memory import SharedMemory

from ..exceptions import UpkieRuntimeError
from .request import Request

_MEBIBYTE = 1 << 20


class AgentInterface:
    def __init__(self, name: str = "/upkie", size: int = 1 * _MEBIBYTE):  # Spine.h:66-69
        self.name = name
        self.size = int(size)
        try:
            self._shm = SharedMemory(name.lstrip("/"), create=True, size=self.size)  # O_CREAT | O_EXCL
        except FileExistsError as exn:  # AgentInterface.cpp:45-54
            raise UpkieRuntimeError(
                f'Cannot open shared memory "{name}": file already exists. Is a spine already running? '
                f"If a previous spine did not exit properly, remove /dev/shm{name if name.startswith('/') else '/' + name}"
            ) from exn
        self._buf = self._shm.buf
        self.set_request(Request.kNone)

    def close(self) -> None:
        if getattr(self, "_shm", None) is not None:
            self._buf = None
            self._shm.close()
            try:
                self._shm.unlink()  # AgentInterface.cpp:81-83
            except FileNotFoundError:
                pass
            self._shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def request(self) -> int:
        return int.from_bytes(self._buf[0:4], sys.byteorder)

    def set_request(self, request: int) -> None:
        self._buf[0:4] = int(request).to_bytes(4, sys.byteorder)

    def data_size(self) -> int:
        return int.from_bytes(self._buf[4:8], sys.byteorder)

    def data(self) -> bytes:
        return bytes(self._buf[8 : 8 + self.data_size()])

    def write(self, data: bytes) -> None:
        if self.size <= len(data) + 8:  # AgentInterface.cpp:91-97
            raise UpkieRuntimeError(
                f"Agent interface buffer overflow: {len(data)} bytes into a buffer of {self.size} bytes"
            )
        self._buf[4:8] = len(data).to_bytes(4, sys.byteorder)
        self._buf[8 : 8 + len(data)] = data
