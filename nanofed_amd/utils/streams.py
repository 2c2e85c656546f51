"""HIP streams for concurrent kernel chains.

PyTorch hands out streams from a pool that it creates in one go, alternating
between its priority levels; the HIP runtime deals hardware queues
(``GPU_MAX_HW_QUEUES``, 4 per process by default) to streams round-robin in
creation order, so consecutive pool streams of one priority can end up sharing
hardware queues - their work then runs one chain after the other although it sits
on different streams.  ``make_streams(n)`` creates ``n`` default-priority streams
back to back with the runtime itself, which spreads them over the hardware queues,
and wraps them for PyTorch; ``destroy_streams`` gives back the ones a caller no
longer needs (the trainer tries several sets and keeps one).
"""

from __future__ import annotations

import ctypes
import os

import torch

_HIP_STREAM_NON_BLOCKING = 0x01
_owned: set[int] = set()  # handles created here and not yet destroyed


def serialized() -> bool:
    """``NANOFED_SERIALIZE_STREAMS=1`` (debug builds of a run): every side stream of the
    engine and the trainers is the current stream, so nothing overlaps and every
    cross-stream dependency is trivially ordered.  A run in this mode must produce the
    same bits as the normal run; a difference means a missing event or wait somewhere
    (``tests/gpu/test_stream_ordering.py``)."""
    return os.environ.get("NANOFED_SERIALIZE_STREAMS", "0") == "1"


def side_stream(device: torch.device) -> torch.cuda.Stream:
    """A stream for work that may overlap the current one (the current stream itself in
    serialized mode)."""
    if serialized():
        return torch.cuda.current_stream(device)
    return torch.cuda.Stream(device=device)


def _runtime() -> ctypes.CDLL:
    """The HIP runtime instance PyTorch itself uses (opening the library by its path
    in this process's mappings, so that no second copy can come in)."""
    try:
        with open("/proc/self/maps") as maps:
            for line in maps:
                path = line.split()[-1]
                if "libamdhip64" in path:
                    return ctypes.CDLL(path)
    except OSError:
        pass
    raise OSError("the HIP runtime is not loaded")


def make_streams(n: int, device: torch.device | int | None = None) -> list[torch.cuda.Stream]:
    """``n`` new default-priority streams on ``device``, created consecutively."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else (
        torch.device(device) if not isinstance(device, int) else torch.device("cuda", device))
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    if serialized():
        return [torch.cuda.current_stream(dev)] * n
    out: list[torch.cuda.Stream] = []
    try:
        hip = _runtime()
        create = hip.hipStreamCreateWithFlags
        create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        create.restype = ctypes.c_int
    except (OSError, AttributeError):
        return [torch.cuda.Stream(dev) for _ in range(n)]
    with torch.cuda.device(index):
        for _ in range(n):
            handle = ctypes.c_void_p()
            if create(ctypes.byref(handle), _HIP_STREAM_NON_BLOCKING) != 0 or not handle.value:
                out.append(torch.cuda.Stream(dev))
                continue
            _owned.add(handle.value)
            out.append(torch.cuda.ExternalStream(handle.value, device=dev))
    return out


def destroy_streams(streams: list[torch.cuda.Stream]) -> int:
    """Destroy streams that ``make_streams`` created (others are left alone: PyTorch
    owns its pool).  The caller must have synchronised with them.  Returns how many
    were destroyed."""
    try:
        destroy = _runtime().hipStreamDestroy
        destroy.argtypes = [ctypes.c_void_p]
        destroy.restype = ctypes.c_int
    except (OSError, AttributeError):
        return 0
    done = 0
    for st in streams:
        handle = int(st.cuda_stream)
        if handle in _owned:
            st.synchronize()
            if destroy(ctypes.c_void_p(handle)) == 0:
                _owned.discard(handle)
                done += 1
    return done
