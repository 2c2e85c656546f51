"""Flat-arena operations: every federation-side pass over a model is one
kernel launch over one contiguous fp32 buffer (csrc/hip/arena_ops.hip).

Each function takes torch tensors as handles.  CUDA(HIP) tensors are
dispatched to the gfx950 kernels and never fall back to eager PyTorch; CPU
tensors run an fp32 PyTorch implementation of the same arithmetic, which is
also the numerical oracle of the kernel tests.

Operations and the reference behaviour they replace:

``weighted_sum``      sum_k w_k * theta_k                (nanofed/server/aggregator/fedavg.py:56-63,
                                                          nanofed/server/aggregator/privacy.py:91-92)
``sqnorm_finite``     squared L2 norm + non-finite count (nanofed/server/validation.py:89-95,111-128)
``clip_noise``        clip to an L2 ball + Gaussian noise (nanofed/privacy/mechanisms.py:85-129,
                                                          nanofed/trainer/private.py:54-86)
``sgd_step``          p -= lr * (g + wd * p)              (torch.optim.SGD in nanofed/trainer/base.py:150)
``philox_fill``       counter-based uniform/normal/laplace (nanofed/privacy/noise/generators.py:49-67)
``scale_accumulate``  out = a * out + b * in              (mask add / unmask, nanofed/server/aggregator/secure.py:230-296)
"""

from __future__ import annotations

import ctypes
import math
from ctypes import c_float, c_int, c_long, c_ulonglong, c_void_p
from typing import Sequence

import torch

from nanofed_amd.core.exceptions import DeviceError

from . import library

MODE_UNIFORM = 0
MODE_NORMAL = 1
MODE_LAPLACE = 2
MODE_SYM_UNIFORM = 3

_workspaces: dict[tuple[int, int], torch.Tensor] = {}


def _workspace(device: torch.device) -> torch.Tensor:
    """Per-(device, stream) scratch for two-stage reductions."""
    key = (device.index or 0, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        n = int(library.hip_lib().nf_reduce_workspace_floats())
        ws = torch.empty(n, dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def _ptr(t: torch.Tensor) -> c_void_p:
    return c_void_p(t.data_ptr())


def _same_flat(tensors: Sequence[torch.Tensor], what: str) -> None:
    first = tensors[0]
    for t in tensors:
        if t.dim() != 1 or t.shape != first.shape or t.device != first.device:
            raise DeviceError(f"{what}: operands must be 1-D, same length, same device")


# ---------------------------------------------------------------------------
def weighted_sum(
    sources: Sequence[torch.Tensor],
    weights: Sequence[float] | torch.Tensor,
    out: torch.Tensor | None = None,
    accumulate: bool = False,
) -> torch.Tensor:
    """``out (+)= sum_k weights[k] * sources[k]`` over flat fp32 buffers.

    ``weights`` may be a device tensor (no host round trip, so the call can be
    recorded in a hipGraph) or a sequence of Python floats.
    """
    if len(sources) == 0:
        raise DeviceError("weighted_sum needs at least one source")
    _same_flat(sources, "weighted_sum")
    dev = sources[0].device
    k, n = len(sources), sources[0].numel()
    if out is None:
        if accumulate:
            raise DeviceError("accumulate=True needs an output buffer")
        out = torch.empty(n, dtype=torch.float32, device=dev)
    if isinstance(weights, torch.Tensor):
        w = weights.to(device=dev, dtype=torch.float32).reshape(-1)
    else:
        w = torch.tensor([float(x) for x in weights], dtype=torch.float32).to(dev)
    if w.numel() != k:
        raise DeviceError("weighted_sum: one weight per source is required")

    if dev.type != "cuda":
        acc = out if accumulate else out.zero_()
        for src, wk in zip(sources, w):
            # zero-weight sources are excluded, not multiplied (they may hold NaNs)
            term = src.to(torch.float32) * wk
            acc.add_(torch.where(wk != 0, term, torch.zeros_like(term)))
        return out

    lib = library.hip_lib()
    cap = int(lib.nf_max_sources())
    for t in (*sources, out):
        library.require_flat_f32(t, "weighted_sum operand")
    first = True
    for lo in range(0, k, cap):
        chunk = sources[lo : lo + cap]
        arr = (c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
        w_chunk = w[lo : lo + cap].contiguous()
        library.check(
            lib.nf_weighted_sum(
                _ptr(out), arr, _ptr(w_chunk), c_int(len(chunk)), c_long(n),
                c_int(1 if (accumulate or not first) else 0),
                library.stream_ptr(dev),
            ),
            "nf_weighted_sum",
        )
        first = False
    return out


# ---------------------------------------------------------------------------
def sqnorm_finite(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """Device tensor ``[sum of squares of the finite entries, number of
    non-finite entries]`` of a flat fp32 buffer, in one pass."""
    if x.dim() != 1:
        raise DeviceError("sqnorm_finite expects a flat buffer")
    dev = x.device
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=dev)
    if dev.type != "cuda":
        xf = x.to(torch.float32)
        ok = torch.isfinite(xf)
        out[0] = torch.where(ok, xf, torch.zeros_like(xf)).square().sum()
        out[1] = (~ok).sum().to(torch.float32)
        return out
    library.require_flat_f32(x, "sqnorm_finite input")
    library.check(
        library.hip_lib().nf_sqnorm_finite(
            _ptr(x), c_long(x.numel()), _ptr(_workspace(dev)), _ptr(out),
            library.stream_ptr(dev),
        ),
        "nf_sqnorm_finite",
    )
    return out


# ---------------------------------------------------------------------------
def clip_noise(
    x: torch.Tensor,
    max_norm: float,
    noise_std: float,
    seed: int,
    offset: int,
    out: torch.Tensor | None = None,
    sqnorm: torch.Tensor | None = None,
) -> torch.Tensor:
    """``out = x * min(1, max_norm / (||x|| + 1e-6)) + N(0, noise_std^2)``.

    The norm is the global L2 norm of the whole buffer (computed here unless
    ``sqnorm`` - a device tensor holding its square - is given).  The noise is
    Philox(seed, offset) indexed by element, so it does not depend on launch
    geometry.  On CPU tensors the same Philox stream is produced by the host
    generator, so CPU and GPU results agree to rounding.
    """
    if x.dim() != 1:
        raise DeviceError("clip_noise expects a flat buffer")
    dev = x.device
    if out is None:
        out = torch.empty_like(x)
    if sqnorm is None:
        sqnorm = sqnorm_finite(x)
    if dev.type != "cuda":
        coef = torch.clamp(max_norm / (sqnorm[0].sqrt() + 1e-6), max=1.0)
        res = x.to(torch.float32) * coef
        if noise_std > 0:
            res = res + noise_std * philox_fill(
                torch.empty_like(res), seed, offset, MODE_NORMAL
            )
        out.copy_(res)
        return out
    library.require_flat_f32(x, "clip_noise input")
    library.require_flat_f32(out, "clip_noise output")
    library.check(
        library.hip_lib().nf_clip_noise(
            _ptr(x), _ptr(out), c_long(x.numel()), _ptr(sqnorm),
            c_float(max_norm), c_float(noise_std),
            c_ulonglong(seed & (2**64 - 1)), c_ulonglong(offset & (2**64 - 1)),
            library.stream_ptr(dev),
        ),
        "nf_clip_noise",
    )
    return out


# ---------------------------------------------------------------------------
def sgd_step(
    params: torch.Tensor,
    grads: torch.Tensor,
    lr: float,
    weight_decay: float = 0.0,
    zero_grad: bool = True,
    bf16_shadow: torch.Tensor | None = None,
) -> None:
    """In-place SGD over a flat parameter buffer and its flat gradient buffer;
    optionally clears the gradients and refreshes a bf16 copy of the
    parameters in the same pass."""
    _same_flat([params, grads], "sgd_step")
    dev = params.device
    if bf16_shadow is not None and (
        bf16_shadow.dtype != torch.bfloat16 or bf16_shadow.numel() != params.numel()
    ):
        raise DeviceError("sgd_step: bf16 shadow must be bfloat16 of the same length")
    if dev.type != "cuda":
        params.sub_(lr * (grads + weight_decay * params))
        if zero_grad:
            grads.zero_()
        if bf16_shadow is not None:
            bf16_shadow.copy_(params)
        return
    library.require_flat_f32(params, "sgd_step params")
    library.require_flat_f32(grads, "sgd_step grads")
    library.check(
        library.hip_lib().nf_sgd_step(
            _ptr(params), _ptr(grads), c_long(params.numel()), c_float(lr),
            c_float(weight_decay), c_int(1 if zero_grad else 0),
            _ptr(bf16_shadow) if bf16_shadow is not None else c_void_p(0),
            library.stream_ptr(dev),
        ),
        "nf_sgd_step",
    )


def sgd_multi(
    params: torch.Tensor,
    grads: list[torch.Tensor],
    offsets: list[int],
    lr: float,
    weight_decay: float = 0.0,
    bf16_shadow: torch.Tensor | None = None,
) -> None:
    """In-place SGD over a flat fp32 parameter buffer whose gradients are separate
    tensors (fp32 or bf16): ``grads[i]``, read in its storage order, updates
    ``params[offsets[i] : offsets[i] + grads[i].numel()]``.  One launch per 96
    tensors; the optional bf16 copy of the parameters is refreshed for the updated
    ranges in the same pass."""
    if len(grads) != len(offsets):
        raise DeviceError("sgd_multi: one offset per gradient tensor")
    dev = params.device
    if bf16_shadow is not None and (
        bf16_shadow.dtype != torch.bfloat16 or bf16_shadow.numel() < params.numel()
    ):
        raise DeviceError("sgd_multi: bf16 shadow must be bfloat16 and cover the parameters")
    for g, off in zip(grads, offsets):
        if g.dtype not in (torch.float32, torch.bfloat16):
            raise DeviceError(f"sgd_multi: gradient dtype {g.dtype} is not supported")
        if off < 0 or off + g.numel() > params.numel():
            raise DeviceError("sgd_multi: gradient range outside the parameter buffer")
    if dev.type != "cuda":
        for g, off in zip(grads, offsets):
            seg = params[off : off + g.numel()]
            flat = _storage_order(g).to(torch.float32)
            seg.sub_(lr * (flat + weight_decay * seg))
            if bf16_shadow is not None:
                bf16_shadow[off : off + g.numel()].copy_(seg)
        return
    library.require_flat_f32(params, "sgd_multi params")
    n = len(grads)
    if n == 0:
        return
    dense = []
    for g in grads:
        if g.device != dev:
            raise DeviceError("sgd_multi: gradients must live on the parameters' device")
        if not _is_dense(g) or g.data_ptr() % 16:
            g = _storage_order(g).clone()
        dense.append(g)
    ptrs = (c_void_p * n)(*[g.data_ptr() for g in dense])
    offs = (c_long * n)(*offsets)
    nums = (c_long * n)(*[g.numel() for g in dense])
    kinds = (c_int * n)(*[1 if g.dtype == torch.bfloat16 else 0 for g in dense])
    library.check(
        library.hip_lib().nf_sgd_multi(
            _ptr(params), _ptr(bf16_shadow) if bf16_shadow is not None else c_void_p(0),
            ptrs, offs, nums, kinds, c_int(n), c_float(lr), c_float(weight_decay),
            library.stream_ptr(dev),
        ),
        "nf_sgd_multi",
    )


def _is_dense(t: torch.Tensor) -> bool:
    """True when the tensor's elements fill one gap-free block of memory."""
    if t.is_contiguous():
        return True
    sizes_strides = sorted(
        ((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1), key=lambda p: p[0]
    )
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True


def _storage_order(t: torch.Tensor) -> torch.Tensor:
    """The elements of a dense tensor as a 1-D tensor in memory order."""
    if t.is_contiguous():
        return t.reshape(-1)
    order = sorted(range(t.dim()), key=lambda d: (-t.stride(d), d))
    return t.permute(order).reshape(-1)


# ---------------------------------------------------------------------------
def _host_philox(
    flat: torch.Tensor, seed: int, offset: int, first: int, mode: int,
    scale: float, accumulate: bool,
) -> None:
    """Host evaluation of the device generator (csrc/host/philox_cpu.cpp)."""
    from nanofed_amd.crypto.backend import _lib as host_lib

    fn = host_lib().nf_philox_cpu
    fn.argtypes = [c_void_p, c_long, c_ulonglong, c_ulonglong, c_long, c_int,
                   c_float, c_int]
    fn.restype = None
    buf = flat if flat.dtype == torch.float32 else flat.to(torch.float32)
    fn(c_void_p(buf.data_ptr()), c_long(buf.numel()),
       c_ulonglong(seed & (2**64 - 1)), c_ulonglong(offset & (2**64 - 1)),
       c_long(first), c_int(mode), c_float(scale), c_int(1 if accumulate else 0))
    if buf is not flat:
        flat.copy_(buf)


def philox_fill(
    out: torch.Tensor,
    seed: int,
    offset: int,
    mode: int = MODE_NORMAL,
    scale: float = 1.0,
    accumulate: bool = False,
    first: int = 0,
) -> torch.Tensor:
    """``out[i] (+)= scale * sample(seed, offset, first + i)``.

    ``first`` is the index of ``out[0]`` in the logical stream, so filling a
    buffer in pieces (or from several GPUs) gives the same values as filling
    it at once.
    """
    flat = out.reshape(-1)
    if flat.data_ptr() != out.data_ptr() or not out.is_contiguous():
        raise DeviceError("philox_fill needs a contiguous output")
    n = flat.numel()
    if n == 0:
        return out
    if out.device.type != "cuda":
        _host_philox(flat, seed, offset, first, mode, scale, accumulate)
        return out
    library.require_flat_f32(flat, "philox_fill output", align=4)
    library.check(
        library.hip_lib().nf_philox_fill(
            _ptr(flat), c_long(n), c_ulonglong(seed & (2**64 - 1)),
            c_ulonglong(offset & (2**64 - 1)), c_long(first), c_int(mode),
            c_float(scale), c_int(1 if accumulate else 0),
            library.stream_ptr(out.device),
        ),
        "nf_philox_fill",
    )
    return out


# ---------------------------------------------------------------------------
def scale_accumulate(out: torch.Tensor, inp: torch.Tensor, a: float, b: float) -> torch.Tensor:
    """``out = a * out + b * inp`` over flat fp32 buffers."""
    _same_flat([out, inp], "scale_accumulate")
    if out.device.type != "cuda":
        out.mul_(a).add_(inp.to(torch.float32), alpha=b)
        return out
    library.require_flat_f32(out, "scale_accumulate out")
    library.require_flat_f32(inp, "scale_accumulate in")
    library.check(
        library.hip_lib().nf_scale_accumulate(
            _ptr(out), _ptr(inp), c_long(out.numel()), c_float(a), c_float(b),
            library.stream_ptr(out.device),
        ),
        "nf_scale_accumulate",
    )
    return out


def l2_norm(x: torch.Tensor) -> float:
    """Host float of the global L2 norm of a flat buffer (one D2H copy)."""
    return math.sqrt(float(sqnorm_finite(x)[0]))
