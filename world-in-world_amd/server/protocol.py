"""Wire protocol of the World-In-World WM server.

Every message is a 4-byte big-endian length followed by `pickle.dumps(obj, HIGHEST_PROTOCOL)`
(reference: downstream/utils/worker_manager.py:249-287 for TCP `read_framed`/`write_framed`,
:213-241 for the pipe variant used between manager and worker).  The literal string "DONE" closes a
session (:589-592, 406-415).  Frames are limited to 4 GiB by the uint32 header.
"""
from __future__ import annotations

import io
import pickle
import struct
import sys

DONE = "DONE"
MAX_FRAME = (1 << 32) - 1


def _loads(data: bytes):
    try:
        return pickle.loads(data)
    except ModuleNotFoundError as e:  # numpy 1.x <-> 2.x pickles name `numpy.core` / `numpy._core` (:266-272)
        if "numpy._core" in str(e) or "numpy.core" in str(e):
            import numpy.core as core  # noqa: F401

            sys.modules.setdefault("numpy._core", sys.modules["numpy.core"])
            for sub in ("multiarray", "numeric", "_multiarray_umath"):
                full = f"numpy.core.{sub}"
                if full in sys.modules:
                    sys.modules.setdefault(f"numpy._core.{sub}", sys.modules[full])
            return pickle.loads(data)
        raise


def dumps_frame(obj) -> bytes:
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    if len(data) > MAX_FRAME:
        raise ValueError(f"frame of {len(data)} bytes exceeds the uint32 length header")
    return struct.pack(">I", len(data)) + data


def _recv_exact(sock, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise EOFError("socket closed " + ("while reading header" if n == 4 else "mid-message"))
        buf += chunk
    return bytes(buf)


def read_framed(sock):
    """TCP side: blocking read of one frame (worker_manager.py:249-272)."""
    (n,) = struct.unpack(">I", _recv_exact(sock, 4))
    return _loads(_recv_exact(sock, n))


def write_framed(sock, obj) -> None:
    """TCP side (worker_manager.py:274-287)."""
    sock.sendall(dumps_frame(obj))


def read_pickled(fileobj: io.BufferedIOBase):
    """Pipe side, blocking (worker_manager.py:213-229).  Raises EOFError on a closed pipe."""
    hdr = b""
    while len(hdr) < 4:
        c = fileobj.read(4 - len(hdr))
        if not c:
            raise EOFError(f"expected 4 bytes, got {len(hdr)} before EOF")
        hdr += c
    n = int.from_bytes(hdr, "big")
    data = bytearray()
    while len(data) < n:
        c = fileobj.read(n - len(data))
        if not c:
            raise EOFError(f"expected {n} bytes, got {len(data)} before EOF")
        data += c
    return _loads(bytes(data))


def write_pickled(fileobj, obj) -> None:
    """Pipe side (worker_manager.py:232-237)."""
    fileobj.write(dumps_frame(obj))
    fileobj.flush()
