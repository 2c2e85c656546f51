"""Loader for the HIP kernel library and shared launch helpers.

``hip_lib()`` returns the ctypes handle of ``libnanofed_hip.so`` (built
in-tree by ``nanofed_amd.native_build``).  On a machine with a GPU the
library is mandatory: device tensors never fall back to eager PyTorch
silently - a missing or unloadable library raises ``DeviceError``.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_ulonglong, c_void_p
from functools import lru_cache

import torch

from nanofed_amd.core.exceptions import DeviceError


@lru_cache(maxsize=1)
def hip_lib() -> ctypes.CDLL:
    from nanofed_amd import native_build

    path = native_build.HIP_LIB
    if not path.exists():
        raise DeviceError(
            f"{path} is missing. Build it with "
            "`python -m nanofed_amd.native_build` (hipcc, gfx950)."
        )
    try:
        lib = ctypes.CDLL(str(path))
    except OSError as exc:
        raise DeviceError(f"Cannot load {path}: {exc}") from exc
    _declare(lib)
    return lib


def _signatures() -> dict:
    P, L, I, F, U = c_void_p, c_long, c_int, c_float, c_ulonglong
    return {
        "nf_max_sources": ([], I),
        "nf_reduce_workspace_floats": ([], I),
        "nf_weighted_sum": ([P, P, P, I, L, I, P], I),
        "nf_sqnorm_finite": ([P, L, P, P, P], I),
        "nf_clip_noise": ([P, P, L, P, F, F, U, U, P], I),
        "nf_sgd_step": ([P, P, L, F, F, I, P, P], I),
        "nf_sgd_multi": ([P, P, P, P, P, P, I, F, F, P], I),
        "nf_philox_fill": ([P, L, U, U, L, I, F, I, P], I),
        "nf_scale_accumulate": ([P, P, L, F, F, P], I),
        "nf_screen_workspace_floats": ([I], I),
        "nf_max_peers": ([], I),
        "nf_screen_rows": ([P, I, L, L, P, P, P, I, I, I, F, P], I),
        "nf_round_weights": ([P, I, I, I, I, I, F, F, I, P, P, P], I),
        "nf_select_copy": ([P, P, P, L, P], I),
        "nf_pairwise_mask": ([P, L, F, P, P, I, F, U, P], I),
        "nf_pairwise_repair": ([P, L, P, P, I, F, U, P], I),
    }


def _declare(lib: ctypes.CDLL) -> None:
    for name, (argtypes, restype) in _signatures().items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue
        fn.argtypes = argtypes
        fn.restype = restype


def hip_lib_symbols_check() -> list[str]:
    """Load the HIP library and make sure every declared launcher is exported
    (the build check: works without a GPU)."""
    lib = hip_lib()
    missing = [n for n in _signatures() if getattr(lib, n, None) is None]
    if missing:
        raise DeviceError(f"libnanofed_hip.so lacks symbols: {missing}")
    return sorted(_signatures())


def declare(name: str, argtypes: list, restype=c_int):
    """Declare (once) and return a function of the HIP library."""
    fn = getattr(hip_lib(), name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


def stream_ptr(device: torch.device | None = None) -> c_void_p:
    """The current HIP stream of ``device`` as a ``hipStream_t``."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


# NANOFED_DEBUG_SYNC=1: wait for the device after every native launch, so that an
# asynchronous error (or a hang) is reported at the launch that caused it instead of at
# some later synchronisation.  Not usable while a stream is being captured into a graph.
_DEBUG_SYNC = os.environ.get("NANOFED_DEBUG_SYNC", "0") == "1"


def check(code: int, what: str) -> None:
    if code != 0:
        raise DeviceError(f"{what} failed with HIP/launch status {code}")
    if _DEBUG_SYNC and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        try:
            torch.cuda.synchronize()
        except RuntimeError as exc:
            raise DeviceError(f"{what}: the device reported an error after the launch: {exc}") from exc


def require_flat_f32(t: torch.Tensor, name: str, align: int = 16) -> None:
    """Host-side operand checks every launcher relies on."""
    if t.dtype != torch.float32:
        raise DeviceError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise DeviceError(f"{name} must be contiguous")
    if t.data_ptr() % align:
        raise DeviceError(f"{name} must be {align}-byte aligned")
