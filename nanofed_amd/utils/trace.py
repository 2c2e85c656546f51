"""Profiler ranges and device-side phase timing.

The reference's only timing facility is ``log_exec`` wall-clock DEBUG lines
(nanofed/utils/logger.py:189-226).  On the device plane two more are needed:

* ``trace_range(name)``: a roctx range (``roctxRangePushA`` / ``roctxRangePop``
  from ``libroctx64.so``), which rocprofv3 shows with ``--marker-trace`` around
  the kernels launched inside it.  Without the library the context manager
  does nothing.
* ``PhaseTimer``: brackets phases of a round with HIP events on the current
  stream.  Recording an event does not synchronise; the elapsed times are read
  after the round's single host synchronisation.  On the CPU it falls back to
  ``time.perf_counter``.
"""

from __future__ import annotations

import ctypes
import time
from contextlib import contextmanager
from typing import Iterator

import torch

_roctx: ctypes.CDLL | None | bool = None


def _load_roctx() -> ctypes.CDLL | None:
    global _roctx
    if _roctx is None:
        _roctx = False
        for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.argtypes = []
                lib.roctxRangePop.restype = ctypes.c_int
                lib.roctxMarkA.argtypes = [ctypes.c_char_p]
                lib.roctxMarkA.restype = None
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
    return _roctx or None


def roctx_available() -> bool:
    return _load_roctx() is not None


@contextmanager
def trace_range(name: str) -> Iterator[None]:
    lib = _load_roctx()
    if lib is None:
        yield
        return
    lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        lib.roctxRangePop()


def trace_mark(name: str) -> None:
    lib = _load_roctx()
    if lib is not None:
        lib.roctxMarkA(name.encode())


class PhaseTimer:
    """``with timer.phase("reduce"): ...`` then ``timer.collect()`` once the
    device work is known to be finished (the caller synchronised)."""

    def __init__(self, device: torch.device | str) -> None:
        self.device = torch.device(device)
        self._gpu = self.device.type == "cuda"
        self._spans: list[tuple[str, object, object]] = []

    @contextmanager
    def phase(self, name: str) -> Iterator[None]:
        with trace_range(f"nanofed.{name}"):
            if self._gpu:
                begin = torch.cuda.Event(enable_timing=True)
                end = torch.cuda.Event(enable_timing=True)
                begin.record()
                try:
                    yield
                finally:
                    end.record()
                    self._spans.append((name, begin, end))
            else:
                t0 = time.perf_counter()
                try:
                    yield
                finally:
                    self._spans.append((name, t0, time.perf_counter()))

    def detach(self) -> list[tuple[str, object, object]]:
        """Hand over the spans recorded so far (for ``summarize``) and start afresh: what is
        recorded from here on belongs to the next collection."""
        spans, self._spans = self._spans, []
        return spans

    def collect(self) -> dict[str, float]:
        """Milliseconds per phase (summed over repeated names); clears the timer."""
        return self.summarize(self.detach())

    def summarize(self, spans: list[tuple[str, object, object]]) -> dict[str, float]:
        out: dict[str, float] = {}
        for name, begin, end in spans:
            if self._gpu:
                end.synchronize()  # type: ignore[union-attr]
                ms = float(begin.elapsed_time(end))  # type: ignore[union-attr]
            else:
                ms = (end - begin) * 1000.0  # type: ignore[operator]
            out[name] = out.get(name, 0.0) + ms
        return out
