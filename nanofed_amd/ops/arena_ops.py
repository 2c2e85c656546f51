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
``screen_rows``       finite / norm verdict of every client row in two launches
                                                         (nanofed/server/validation.py:84-101 per update)
``round_weights``     z-score anomaly screen + n_k / sum(n) weights + quorum flag on device
                                                         (nanofed/server/validation.py:103-135,
                                                          nanofed/server/aggregator/fedavg.py:101-125,
                                                          nanofed/orchestration/coordinator.py:209-236)
``select_copy``       dst = src when a device flag is set (the old model survives a failed round)
``pairwise_mask`` / ``pairwise_repair``  seeded pairwise blinding of a row, removal of the
                      masks a dropped client left unmatched (nanofed/server/aggregator/secure.py:192-313)
"""

from __future__ import annotations

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


# ---------------------------------------------------------------------------
_screen_ws: dict[tuple[int, int, int], torch.Tensor] = {}


def screen_rows(
    rows: torch.Tensor,
    records: torch.Tensor,
    col_sq: int,
    col_valid: int,
    ref: torch.Tensor | None = None,
    max_norm: float = 0.0,
) -> torch.Tensor:
    """Screen the G client rows of a pool (``rows`` is ``[G, n]``, rows may be
    strided) in two launches: ``records[g, col_sq]`` receives the squared L2 norm
    of row g (of ``row - ref`` when ``ref`` is given) over its finite entries and
    ``records[g, col_valid]`` is zeroed when the row holds a non-finite value or,
    with ``max_norm > 0``, its norm exceeds the bound."""
    if rows.dim() != 2 or records.dim() != 2 or records.shape[0] < rows.shape[0]:
        raise DeviceError("screen_rows: rows [G, n] and records [>=G, M] are required")
    G, n = rows.shape
    dev = rows.device
    max_sq = float(max_norm) ** 2 if max_norm > 0 else 0.0
    if dev.type != "cuda":
        x = rows.to(torch.float32)
        if ref is not None:
            x = x - ref.to(torch.float32)
        ok = torch.isfinite(x)
        sq = torch.where(ok, x, torch.zeros_like(x)).square().sum(dim=1)
        valid = ok.all(dim=1)
        if max_sq > 0:
            valid = valid & (sq <= max_sq)
        records[:G, col_sq] = sq
        records[:G, col_valid] = records[:G, col_valid] * valid.to(records.dtype)
        return records
    if rows.dtype != torch.float32 or rows.stride(1) != 1 or rows.data_ptr() % 16:
        raise DeviceError("screen_rows: rows must be float32, unit inner stride, 16-byte aligned")
    # the kernel reads float4 pieces: every row must start on a 16-byte boundary
    row_stride = rows.stride(0) if G > 1 else (n + 3) & ~3
    if row_stride % 4:
        raise DeviceError("screen_rows: the row stride must be a multiple of 4 floats")
    library.require_flat_f32(records, "screen_rows records", align=4)
    if ref is not None:
        library.require_flat_f32(ref, "screen_rows reference")
        if ref.numel() != n:
            raise DeviceError("screen_rows: reference length differs from the rows")
    lib = library.hip_lib()
    key = (dev.index or 0, torch.cuda.current_stream(dev).cuda_stream, G)
    ws = _screen_ws.get(key)
    if ws is None:
        ws = torch.empty(int(lib.nf_screen_workspace_floats(G)), dtype=torch.float32, device=dev)
        _screen_ws[key] = ws
    library.check(
        lib.nf_screen_rows(
            _ptr(rows), c_int(G), c_long(row_stride), c_long(n),
            _ptr(ref) if ref is not None else c_void_p(0), _ptr(ws), _ptr(records),
            c_int(records.stride(0)), c_int(col_sq), c_int(col_valid), c_float(max_sq),
            library.stream_ptr(dev),
        ),
        "nf_screen_rows",
    )
    return records


def round_weights(
    records: torch.Tensor,
    col_samples: int,
    col_valid: int,
    col_sq: int,
    required: float,
    weights: torch.Tensor,
    flags: torch.Tensor,
    z_threshold: float = 0.0,
    min_clients_for_stats: int = 5,
) -> None:
    """From the gathered ``[K, M]`` client records, on the device: optional
    leave-one-out z-score screen of the update norms (a client more than
    ``z_threshold`` standard deviations from the mean of the other valid clients
    is marked invalid in ``records``), ``weights[k] = valid_k n_k / sum(valid n)``
    and ``flags = [round ok, valid clients, total samples, anomalous clients]``."""
    K, M = records.shape
    if weights.numel() < K or flags.numel() < 4:
        raise DeviceError("round_weights: weights [K] and flags [4] are required")
    dev = records.device
    if dev.type != "cuda":
        valid = records[:, col_valid] > 0
        anomalous = 0
        if z_threshold > 0:
            x = records[:, col_sq].to(torch.float64).sqrt()
            m = int(valid.sum())
            others = m - 1
            if others >= min_clients_for_stats and others >= 2:
                S = x[valid].sum()
                Q = x[valid].square().sum()
                mean = (S - x) / others
                var = ((Q - x * x - others * mean * mean) / (others - 1)).clamp(min=0.0)
                z = (x - mean).abs() / (var.sqrt() + 1e-8)
                bad = valid & (z > z_threshold)
                anomalous = int(bad.sum())
                records[:, col_valid] = torch.where(
                    bad, torch.zeros_like(records[:, col_valid]), records[:, col_valid]
                )
                valid = records[:, col_valid] > 0
        v = valid.to(torch.float32)
        mass = v * records[:, col_samples]
        total = mass.sum()
        weights[:K] = mass / total.clamp(min=1e-30)
        flags[0] = float(bool(v.sum() >= required) and bool(total > 0))
        flags[1] = v.sum()
        flags[2] = total
        flags[3] = float(anomalous)
        return
    library.require_flat_f32(records, "round_weights records", align=4)
    library.require_flat_f32(weights, "round_weights weights", align=4)
    library.require_flat_f32(flags, "round_weights flags", align=4)
    library.check(
        library.hip_lib().nf_round_weights(
            _ptr(records), c_int(K), c_int(M), c_int(col_samples), c_int(col_valid),
            c_int(col_sq), c_float(required), c_float(z_threshold),
            c_int(min_clients_for_stats), _ptr(weights), _ptr(flags), library.stream_ptr(dev),
        ),
        "nf_round_weights",
    )


def select_copy(dst: torch.Tensor, src: torch.Tensor, flag: torch.Tensor) -> torch.Tensor:
    """``dst = src`` when ``flag[0] > 0`` (decided on the device)."""
    _same_flat([dst, src], "select_copy")
    if dst.device.type != "cuda":
        if bool(flag.reshape(-1)[0] > 0):
            dst.copy_(src)
        return dst
    library.require_flat_f32(dst, "select_copy dst")
    library.require_flat_f32(src, "select_copy src")
    library.check(
        library.hip_lib().nf_select_copy(
            _ptr(dst), _ptr(src), _ptr(flag), c_long(dst.numel()),
            library.stream_ptr(dst.device),
        ),
        "nf_select_copy",
    )
    return dst


class PeerMaskPlan:
    """A client's pair seeds and signs, marshalled once for the launcher: the per-round
    call then costs one launch per ``nf_max_peers`` peers and no Python loop over them."""

    def __init__(self, seeds: Sequence[int], signs: Sequence[float]) -> None:
        if len(seeds) != len(signs):
            raise DeviceError("PeerMaskPlan: one sign per seed is required")
        self.seeds = [int(s) for s in seeds]
        self.signs = [float(v) for v in signs]
        self._chunks: list[tuple] | None = None

    def chunks(self) -> list[tuple]:
        if self._chunks is None:
            cap = int(library.hip_lib().nf_max_peers())
            self._chunks = []
            for lo in range(0, max(len(self.seeds), 1), cap):
                chunk, sg = self.seeds[lo : lo + cap], self.signs[lo : lo + cap]
                s_arr = (c_ulonglong * max(len(chunk), 1))(*[s & (2**64 - 1) for s in chunk])
                f_arr = (c_float * max(len(chunk), 1))(*sg)
                self._chunks.append((s_arr, f_arr, len(chunk)))
        return self._chunks


def pairwise_mask_(
    row: torch.Tensor,
    seeds: "Sequence[int] | PeerMaskPlan",
    signs: Sequence[float] | None,
    scale: float,
    prescale: float = 1.0,
    offset: int = 0,
) -> torch.Tensor:
    """``row = prescale * row + sum_p signs[p] * scale * U_p`` where ``U_p`` is the
    symmetric-uniform Philox stream of ``seeds[p]`` (what ``mask_add_`` draws): a
    client's blinding against all its peers in one pass.  ``seeds`` may be a
    ``PeerMaskPlan`` (then ``signs`` is ignored)."""
    plan = seeds if isinstance(seeds, PeerMaskPlan) else PeerMaskPlan(seeds, signs or [])
    if row.dim() != 1:
        raise DeviceError("pairwise_mask_: a flat row is required")
    if row.device.type != "cuda":
        row.mul_(prescale)
        for seed, sign in zip(plan.seeds, plan.signs):
            philox_fill(row, seed, offset, MODE_SYM_UNIFORM, scale=sign * scale, accumulate=True)
        return row
    library.require_flat_f32(row, "pairwise_mask_ row")
    lib = library.hip_lib()
    first = True
    for s_arr, f_arr, count in plan.chunks():
        library.check(
            lib.nf_pairwise_mask(
                _ptr(row), c_long(row.numel()), c_float(prescale if first else 1.0), s_arr,
                f_arr, c_int(count), c_float(scale), c_ulonglong(offset & (2**64 - 1)),
                library.stream_ptr(row.device),
            ),
            "nf_pairwise_mask",
        )
        first = False
    return row


def pairwise_repair_(
    total: torch.Tensor,
    present: torch.Tensor,
    seeds: torch.Tensor,
    scale: float,
    offset: int = 0,
) -> torch.Tensor:
    """Remove from a sum of blinded rows the masks that found no partner: for
    every pair with exactly one present client, subtract what the survivor added.
    ``present`` is ``[K]`` floats and ``seeds`` ``[K, K]`` int64 on ``total``'s
    device, so the decision needs no host round trip."""
    K = present.numel()
    if seeds.shape != (K, K) or seeds.dtype != torch.int64:
        raise DeviceError("pairwise_repair_: seeds must be a [K, K] int64 table")
    if total.device.type != "cuda":
        pres = [bool(v > 0) for v in present.tolist()]
        table = seeds.tolist()
        for i in range(K):
            for j in range(i + 1, K):
                if pres[i] == pres[j]:
                    continue
                a = scale if pres[i] else -scale
                philox_fill(total, table[i][j], offset, MODE_SYM_UNIFORM, scale=-a,
                            accumulate=True)
        return total
    library.require_flat_f32(total, "pairwise_repair_ total")
    library.require_flat_f32(present, "pairwise_repair_ present", align=4)
    if not seeds.is_contiguous():
        raise DeviceError("pairwise_repair_: seeds must be contiguous")
    library.check(
        library.hip_lib().nf_pairwise_repair(
            _ptr(total), c_long(total.numel()), _ptr(present), _ptr(seeds), c_int(K),
            c_float(scale), c_ulonglong(offset & (2**64 - 1)), library.stream_ptr(total.device),
        ),
        "nf_pairwise_repair",
    )
    return total


def l2_norm(x: torch.Tensor) -> float:
    """Host float of the global L2 norm of a flat buffer (one D2H copy)."""
    return math.sqrt(float(sqnorm_finite(x)[0]))
