"""Client-grouped launches: one kernel call steps a layer for ALL clients of a GPU.

The batch holds ``G`` groups (clients) of ``B`` images, one after the other, and every
group has parameters of its own: group ``g``'s tensor starts a fixed number of
elements (a row of the parameter arena, of its bf16 copy, or of the flat gradient
buffer) after group ``g - 1``'s.  Parameters are therefore passed as ``Ref(base,
offset)`` - a flat ``[G, row]`` tensor and the element offset of the tensor inside a
row - never as per-client tensor lists.

Kernels: csrc/hip/conv_ops.hip (``*_g`` entry points), bn_ops.hip, gemm_ops.hip
(``nf_gemm_bf16_batched``), im2col_ops.hip, resnet_ops.hip.  The reference steps one
client per process (nanofed/trainer/base.py:141-159); these ops serve
``nanofed_amd.engine.resnet_native``.
"""

from __future__ import annotations

from ctypes import c_float, c_int, c_long, c_void_p
from dataclasses import dataclass

import torch

from nanofed_amd.core.exceptions import DeviceError

from . import gemm_ops, library

P, L, I, FL = c_void_p, c_long, c_int, c_float
_SIGS = {
    "nf_conv3x3_supported_g": ([I, I, I, I, I, I], I),
    "nf_conv3x3_fwd_tiles_g": ([I, I, I, I, I], I),
    "nf_conv3x3_fwd_g": ([P, P, P, P, I, I, I, I, I, I, L, P], I),
    "nf_conv3x3_dgrad_g": ([P, P, P, I, I, I, I, I, I, L, P], I),
    "nf_conv3x3_dgrad_bn_g": ([P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, L, P], I),
    "nf_conv3x3_wgrad_workspace_g": ([I, I, I, I, I, I], L),
    "nf_conv3x3_wgrad_g": ([P, P, P, P, I, I, I, I, I, I, L, P], I),
    "nf_bn_fwd_g": ([P, P, P, P, P, P, P, P, P, I, L, I, I, FL, FL, I, L, P], I),
    "nf_bn_bwd_g": ([P, P, P, P, P, P, P, P, P, P, P, I, L, I, I, L, I, L, P], I),
    "nf_bn_bwd_pre_g": ([P, P, P, P, P, P, P, P, P, I, L, I, I, L, I, L, P], I),
    "nf_bn_workspace_floats": ([I], I),
    "nf_gemm_bf16_batched": ([I, I, P, P, P, P, P, P, I, I, I, L, L, L, I, I, L, L, L, L, P], I),
    "nf_im2col": ([P, P, I, I, I, I, I, I, I, I, I, I, I, P], I),
    "nf_col2im": ([P, P, I, I, I, I, I, I, I, I, I, I, I, P], I),
    "nf_col2im_join_blocks": ([I, I, I, I], I),
    "nf_col2im_join_g": ([P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P], I),
    "nf_stem_patches": ([P, P, P, I, I, I, P, P, P], I),
    "nf_pool_fc_ce": ([P, P, P, P, P, P, P, P, P, I, I, I, I, I, L, L, I, P], I),
    "nf_fc_grads": ([P, P, P, P, P, P, P, I, I, I, I, L, P], I),
    "nf_sgd_rows": ([P, P, P, I, L, L, L, L, FL, FL, P], I),
}
_fns: dict[str, object] = {}


def _fn(name: str):
    fn = _fns.get(name)
    if fn is None:
        argtypes, restype = _SIGS[name]
        fn = library.declare(name, argtypes, restype)
        _fns[name] = fn
    return fn


def symbols() -> list[str]:
    return sorted(_SIGS)


def _p(t: torch.Tensor | None) -> c_void_p:
    return c_void_p(t.data_ptr() if t is not None else 0)


@dataclass(frozen=True)
class Ref:
    """A per-group tensor inside a flat ``[G, row]`` buffer: group 0's copy starts
    ``offset`` elements into ``base``'s first row, group g's ``g * stride`` later."""

    base: torch.Tensor
    offset: int

    @property
    def ptr(self) -> c_void_p:
        return c_void_p(self.base.data_ptr() + self.offset * self.base.element_size())

    @property
    def stride(self) -> int:
        return int(self.base.stride(0))

    def view(self, g: int, shape: tuple[int, ...]) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        return self.base[g, self.offset : self.offset + n].view(shape)


def _stream(t: torch.Tensor) -> c_void_p:
    return library.stream_ptr(t.device)


# Host-side checks.  The launchers take raw addresses plus the extents the caller says they
# have; a slip in a caller would be an out-of-bounds device access (which can take the whole
# GPU down), so every operand is checked here for device, dtype, contiguity and size.
# (a tuple so that the CPU unit tests of these checks can admit host tensors: every case
# they try is rejected before a launch)
_DEVICE_TYPES: tuple[str, ...] = ("cuda",)


def _need(t: torch.Tensor | None, what: str, dtype: torch.dtype | tuple, numel: int,
          optional: bool = False) -> None:
    if t is None:
        if optional:
            return
        raise DeviceError(f"{what}: missing")
    dtypes = dtype if isinstance(dtype, tuple) else (dtype,)
    if t.device.type not in _DEVICE_TYPES:
        raise DeviceError(f"{what}: a device tensor is required, got {t.device}")
    if t.dtype not in dtypes:
        raise DeviceError(f"{what}: dtype {t.dtype}, expected {' or '.join(map(str, dtypes))}")
    dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    if not dense:
        raise DeviceError(f"{what}: a dense tensor (row-major or channels-last) is required")
    if t.numel() < numel:
        raise DeviceError(f"{what}: {t.numel()} elements, the launch touches {numel}")


def _need_ref(r: "Ref | None", what: str, dtype: torch.dtype | tuple, numel: int, G: int,
              optional: bool = False) -> None:
    """Group g's tensor = ``numel`` elements at ``offset`` of row g of ``base``."""
    if r is None:
        if optional:
            return
        raise DeviceError(f"{what}: missing")
    base = r.base
    dtypes = dtype if isinstance(dtype, tuple) else (dtype,)
    if (base.device.type not in _DEVICE_TYPES or base.dtype not in dtypes or base.dim() != 2
            or base.stride(1) != 1):
        raise DeviceError(f"{what}: a [groups, row] device buffer of {dtypes} is required")
    if base.shape[0] < G:
        raise DeviceError(f"{what}: {base.shape[0]} rows for {G} groups")
    if r.offset < 0 or r.offset + numel > base.shape[1]:
        raise DeviceError(f"{what}: elements {r.offset}..{r.offset + numel} lie outside a row of {base.shape[1]}")


def _positive(**dims: int) -> None:
    for name, v in dims.items():
        if int(v) < 1:
            raise DeviceError(f"{name} = {v}: a positive extent is required")


BF16, F32 = torch.bfloat16, torch.float32


# -- 3x3 / stride 1 convolution ---------------------------------------------------------

def conv3x3_supported(G: int, B: int, H: int, W: int, c_in: int, c_out: int) -> bool:
    return bool(_fn("nf_conv3x3_supported_g")(G, B, H, W, c_in, c_out))


def conv3x3_tiles(G: int, B: int, H: int, W: int, c_out: int) -> int:
    """Rows of per-channel partial sums the forward launch writes per group."""
    return int(_fn("nf_conv3x3_fwd_tiles_g")(G, B, H, W, c_out))


def conv3x3_wgrad_workspace(G: int, B: int, H: int, W: int, c_in: int, c_out: int) -> int:
    n = int(_fn("nf_conv3x3_wgrad_workspace_g")(G, B, H, W, c_in, c_out))
    if n <= 0:
        raise DeviceError("grouped conv3x3 weight gradient: unsupported shape")
    return n


def conv3x3_fwd(x: torch.Tensor, w: Ref, y: torch.Tensor, colsum: torch.Tensor | None, G: int,
                B: int, H: int, W: int, c_in: int, c_out: int) -> None:
    """x ``[G * B, H, W, c_in]`` -> y ``[G * B, H, W, c_out]`` (bf16, channels-last
    storage); ``colsum`` ``[G * tiles, 2, c_out]`` float32 receives per-tile sums of y
    and y^2."""
    _positive(G=G, B=B, H=H, W=W, c_in=c_in, c_out=c_out)
    if not conv3x3_supported(G, B, H, W, c_in, c_out):
        raise DeviceError(f"conv3x3_fwd: shape G={G} B={B} {H}x{W} {c_in}->{c_out} is not covered")
    _need(x, "conv3x3_fwd x", BF16, G * B * H * W * c_in)
    _need(y, "conv3x3_fwd y", BF16, G * B * H * W * c_out)
    _need_ref(w, "conv3x3_fwd w", BF16, 9 * c_in * c_out, G)
    if colsum is not None:
        _need(colsum, "conv3x3_fwd colsum", F32, G * conv3x3_tiles(G, B, H, W, c_out) * 2 * c_out)
    library.check(
        _fn("nf_conv3x3_fwd_g")(_p(x), w.ptr, _p(y), _p(colsum), G, B, H, W, c_in, c_out,
                                w.stride, _stream(x)),
        "conv3x3_fwd_g",
    )


@dataclass(frozen=True)
class BnBwdFuse:
    """What the data gradient needs to do the elementwise part of the backward pass of the
    BatchNorm that produced the convolution's input: ``mask`` (the ReLU output after that
    BatchNorm = the convolution's input), ``xpre`` (the BatchNorm's input), ``add`` (the
    gradient from the second branch that read the ReLU output, or None), all of the shape
    of ``dx``, and ``colsum`` ``[G * conv3x3_tiles(G, B, H, W, c_in), 2, c_in]`` float32 for
    the per-tile sums of the masked gradient and of its product with ``xpre``.

    ``mask=None`` with ``stats`` ``[G, 2, c_in]``, ``gamma`` and ``beta`` (float32 ``Ref``s):
    the BatchNorm had no residual input, and the kernel recomputes the sign of its output
    from ``xpre`` instead of reading the output."""

    mask: torch.Tensor | None
    xpre: torch.Tensor
    add: torch.Tensor | None
    colsum: torch.Tensor
    stats: torch.Tensor | None = None
    gamma: Ref | None = None
    beta: Ref | None = None


def conv3x3_dgrad(dy: torch.Tensor, w: Ref, dx: torch.Tensor, G: int, B: int, H: int, W: int,
                  c_in: int, c_out: int, fuse: BnBwdFuse | None = None) -> None:
    """dx = data gradient; with ``fuse``: dx = (data gradient + add) where mask > 0, else 0,
    and the partial sums ``bn_bwd_pre`` takes."""
    _positive(G=G, B=B, H=H, W=W, c_in=c_in, c_out=c_out)
    # (the data gradient is the forward kernel run on dy: its "input" width is c_out)
    if not conv3x3_supported(G, B, H, W, c_out, c_in):
        raise DeviceError(f"conv3x3_dgrad: shape G={G} B={B} {H}x{W} {c_in}->{c_out} is not covered")
    n_in = G * B * H * W * c_in
    _need(dy, "conv3x3_dgrad dy", BF16, G * B * H * W * c_out)
    _need(dx, "conv3x3_dgrad dx", BF16, n_in)
    _need_ref(w, "conv3x3_dgrad w", BF16, 9 * c_in * c_out, G)
    if fuse is None:
        library.check(
            _fn("nf_conv3x3_dgrad_g")(_p(dy), w.ptr, _p(dx), G, B, H, W, c_in, c_out, w.stride,
                                      _stream(dy)),
            "conv3x3_dgrad_g",
        )
        return
    null, pstride = c_void_p(0), 0
    stats = gamma = beta = null
    if fuse.mask is None:
        _need(fuse.stats, "conv3x3_dgrad stats", F32, G * 2 * c_in)
        _need_ref(fuse.gamma, "conv3x3_dgrad gamma", F32, c_in, G)
        _need_ref(fuse.beta, "conv3x3_dgrad beta", F32, c_in, G)
        if fuse.gamma.stride != fuse.beta.stride:
            raise DeviceError("conv3x3_dgrad: gamma and beta must be rows of one arena")
        stats, gamma, beta, pstride = _p(fuse.stats), fuse.gamma.ptr, fuse.beta.ptr, fuse.gamma.stride
    else:
        _need(fuse.mask, "conv3x3_dgrad mask", BF16, n_in)
    _need(fuse.xpre, "conv3x3_dgrad xpre", BF16, n_in)
    _need(fuse.add, "conv3x3_dgrad add", BF16, n_in, optional=True)
    _need(fuse.colsum, "conv3x3_dgrad colsum", F32, G * conv3x3_tiles(G, B, H, W, c_in) * 2 * c_in)
    for t, what in ((fuse.mask, "mask"), (fuse.xpre, "xpre"), (fuse.add, "add")):
        if t is not None and t.data_ptr() == dx.data_ptr():
            raise DeviceError(f"conv3x3_dgrad: dx must not alias {what}")
    library.check(
        _fn("nf_conv3x3_dgrad_bn_g")(_p(dy), w.ptr, _p(dx), _p(fuse.mask), _p(fuse.xpre),
                                     _p(fuse.add), _p(fuse.colsum), stats, gamma, beta, pstride,
                                     G, B, H, W, c_in, c_out, w.stride, _stream(dy)),
        "conv3x3_dgrad_bn_g",
    )


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: Ref, workspace: torch.Tensor, G: int,
                  B: int, H: int, W: int, c_in: int, c_out: int) -> None:
    _positive(G=G, B=B, H=H, W=W, c_in=c_in, c_out=c_out)
    _need(x, "conv3x3_wgrad x", BF16, G * B * H * W * c_in)
    _need(dy, "conv3x3_wgrad dy", BF16, G * B * H * W * c_out)
    _need_ref(dw, "conv3x3_wgrad dw", (BF16, F32), 9 * c_in * c_out, G)
    _need(workspace, "conv3x3_wgrad workspace", (BF16, F32),
          conv3x3_wgrad_workspace(G, B, H, W, c_in, c_out))
    library.check(
        _fn("nf_conv3x3_wgrad_g")(_p(x), _p(dy), dw.ptr, _p(workspace), G, B, H, W, c_in, c_out,
                                  dw.stride, _stream(x)),
        "conv3x3_wgrad_g",
    )


# -- BatchNorm (+ residual) (+ ReLU) ------------------------------------------------------

def bn_workspace_floats(G: int, C: int) -> int:
    """Floats of workspace the grouped forward / backward launches need."""
    return G * (int(_fn("nf_bn_workspace_floats")(C)) + 2 * C)


def bn_fwd(x: torch.Tensor, res: torch.Tensor | None, y: torch.Tensor, gamma: Ref, beta: Ref,
           running_mean: Ref | None, running_var: Ref | None, stats: torch.Tensor,
           workspace: torch.Tensor, G: int, rows: int, C: int, relu: bool, momentum: float,
           eps: float, pre_blocks: int = 0) -> None:
    """Training-mode BatchNorm of G groups of ``rows`` pixels.  gamma / beta / running
    statistics are float32 ``Ref``s into the parameter arena; ``stats`` ``[G, 2, C]``
    keeps mean and 1/std for the backward pass.  ``pre_blocks`` > 0: ``workspace`` holds
    ``[G, pre_blocks, 2, C]`` partial sums written by the convolution."""
    _positive(G=G, rows=rows, C=C)
    if C % 8:
        raise DeviceError(f"bn_fwd: {C} channels (a multiple of 8 is required)")
    _need(x, "bn_fwd x", BF16, G * rows * C)
    _need(res, "bn_fwd residual", BF16, G * rows * C, optional=True)
    _need(y, "bn_fwd y", BF16, G * rows * C)
    for r, what in ((gamma, "gamma"), (beta, "beta")):
        _need_ref(r, f"bn_fwd {what}", F32, C, G)
    for r, what in ((running_mean, "running_mean"), (running_var, "running_var")):
        _need_ref(r, f"bn_fwd {what}", F32, C, G, optional=True)
    if beta.stride != gamma.stride or (running_mean is not None and running_mean.stride != gamma.stride):
        raise DeviceError("bn_fwd: the parameter vectors must be rows of one arena")
    _need(stats, "bn_fwd stats", F32, G * 2 * C)
    _need(workspace, "bn_fwd workspace", F32,
          max(bn_workspace_floats(G, C), G * pre_blocks * 2 * C))
    library.check(
        _fn("nf_bn_fwd_g")(_p(x), _p(res), _p(y), gamma.ptr, beta.ptr,
                           running_mean.ptr if running_mean is not None else c_void_p(0),
                           running_var.ptr if running_var is not None else c_void_p(0),
                           _p(stats), _p(workspace), G, rows, C, int(relu), momentum, eps,
                           pre_blocks, gamma.stride, _stream(x)),
        "bn_fwd_g",
    )


def bn_bwd(dy: torch.Tensor, dy2: torch.Tensor | None, y: torch.Tensor | None, x: torch.Tensor,
           stats: torch.Tensor, gamma: Ref, dx: torch.Tensor, dres: torch.Tensor | None,
           dgamma: Ref, dbeta: Ref, workspace: torch.Tensor, G: int, rows: int, C: int,
           relu: bool) -> None:
    """Backward of ``bn_fwd``.  The incoming gradient is ``dy (+ dy2)``; ``dgamma`` /
    ``dbeta`` are ``Ref``s into a bf16 or float32 gradient buffer."""
    if dgamma.base.dtype != dbeta.base.dtype or dgamma.stride != dbeta.stride:
        raise DeviceError("bn_bwd: d(gamma) and d(beta) must live in the same buffer")
    out_bf16 = dgamma.base.dtype == torch.bfloat16
    _positive(G=G, rows=rows, C=C)
    if C % 8:
        raise DeviceError(f"bn_bwd: {C} channels (a multiple of 8 is required)")
    _need(dy, "bn_bwd dy", BF16, G * rows * C)
    _need(dy2, "bn_bwd dy2", BF16, G * rows * C, optional=True)
    if relu:
        _need(y, "bn_bwd y", BF16, G * rows * C)
    _need(x, "bn_bwd x", BF16, G * rows * C)
    _need(stats, "bn_bwd stats", F32, G * 2 * C)
    _need_ref(gamma, "bn_bwd gamma", F32, C, G)
    _need(dx, "bn_bwd dx", BF16, G * rows * C)
    _need(dres, "bn_bwd d(residual)", BF16, G * rows * C, optional=True)
    _need_ref(dgamma, "bn_bwd d(gamma)", (BF16, F32), C, G)
    _need_ref(dbeta, "bn_bwd d(beta)", (BF16, F32), C, G)
    _need(workspace, "bn_bwd workspace", F32, bn_workspace_floats(G, C))
    library.check(
        _fn("nf_bn_bwd_g")(_p(dy), _p(dy2), _p(y if relu else None), _p(x), _p(stats), gamma.ptr,
                           _p(dx), _p(dres), dgamma.ptr, dbeta.ptr, _p(workspace), G, rows, C,
                           int(relu), gamma.stride, int(out_bf16), dgamma.stride, _stream(x)),
        "bn_bwd_g",
    )


def bn_bwd_pre(dz: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, gamma: Ref,
               dx: torch.Tensor, dgamma: Ref, dbeta: Ref, pre: torch.Tensor, pre_blocks: int,
               workspace: torch.Tensor, G: int, rows: int, C: int) -> None:
    """Backward of ``bn_fwd`` for an incoming gradient ``dz`` that is final (ReLU mask
    applied, branches added) and whose partial sums ``pre`` ``[G, pre_blocks, 2, C]`` (of dz
    and dz * x over disjoint pixel ranges) were written by ``conv3x3_dgrad(fuse=...)``:
    the fixed-order final sum and one pass that reads dz and x and writes dx."""
    if dgamma.base.dtype != dbeta.base.dtype or dgamma.stride != dbeta.stride:
        raise DeviceError("bn_bwd_pre: d(gamma) and d(beta) must live in the same buffer")
    out_bf16 = dgamma.base.dtype == torch.bfloat16
    _positive(G=G, rows=rows, C=C, pre_blocks=pre_blocks)
    if C % 8:
        raise DeviceError(f"bn_bwd_pre: {C} channels (a multiple of 8 is required)")
    _need(dz, "bn_bwd_pre dz", BF16, G * rows * C)
    _need(x, "bn_bwd_pre x", BF16, G * rows * C)
    _need(stats, "bn_bwd_pre stats", F32, G * 2 * C)
    _need_ref(gamma, "bn_bwd_pre gamma", F32, C, G)
    _need(dx, "bn_bwd_pre dx", BF16, G * rows * C)
    _need_ref(dgamma, "bn_bwd_pre d(gamma)", (BF16, F32), C, G)
    _need_ref(dbeta, "bn_bwd_pre d(beta)", (BF16, F32), C, G)
    _need(pre, "bn_bwd_pre partial sums", F32, G * pre_blocks * 2 * C)
    _need(workspace, "bn_bwd_pre workspace", F32, bn_workspace_floats(G, C))
    library.check(
        _fn("nf_bn_bwd_pre_g")(_p(dz), _p(x), _p(stats), gamma.ptr, _p(dx), dgamma.ptr, dbeta.ptr,
                               _p(pre), _p(workspace), G, rows, C, pre_blocks, gamma.stride,
                               int(out_bf16), dgamma.stride, _stream(x)),
        "bn_bwd_pre_g",
    )


# -- batched GEMM -------------------------------------------------------------------------

def gemm_slab_floats(M: int, N: int, K: int, layout: int, batch: int) -> int:
    geo = gemm_ops.choose_geometry(M, N, K, layout, batch)
    splits = gemm_ops.choose_splits(M, N, K, geo, batch)
    return batch * splits * M * N if splits > 1 else 0


Operand = "torch.Tensor | Ref"


def _span(x, what: str) -> tuple[int, int]:
    """(address, bf16 elements available from there to the end of the buffer) of a GEMM
    operand given as a tensor or as a ``Ref`` into a ``[groups, row]`` buffer."""
    if isinstance(x, Ref):
        base = x.base
        if (base.device.type not in _DEVICE_TYPES or base.dtype != BF16 or base.dim() != 2
                or base.stride(1) != 1):
            raise DeviceError(f"{what}: a [groups, row] bf16 device buffer is required")
        if not 0 <= x.offset <= base.shape[1]:
            raise DeviceError(f"{what}: offset {x.offset} outside a row of {base.shape[1]}")
        avail = (base.shape[0] - 1) * base.stride(0) + base.shape[1] - x.offset
        return x.ptr.value, avail
    if not isinstance(x, torch.Tensor):
        raise DeviceError(f"{what}: a tensor or a Ref is required (raw addresses cannot be checked)")
    if x.device.type not in _DEVICE_TYPES or x.dtype != BF16:
        raise DeviceError(f"{what}: a bf16 device tensor is required")
    store = x.untyped_storage().nbytes() // x.element_size() - x.storage_offset()
    return x.data_ptr(), store


def _extent(rows: int, cols: int, ld: int, batch: int, stride: int) -> int:
    """Elements from the first to one past the last element of ``batch`` row-major
    ``rows x cols`` matrices with row pitch ``ld``, ``stride`` apart."""
    return (batch - 1) * stride + (rows - 1) * ld + cols


def gemm_batched(layout: int, a: Operand, b: Operand, c: Operand, M: int, N: int, K: int,
                 lda: int, ldb: int, ldc: int, batch: int, sa: int, sb: int, sc: int,
                 slabs: torch.Tensor | None, stream: c_void_p, bias: Ref | None = None,
                 c2: torch.Tensor | None = None) -> None:
    """``batch`` GEMMs of one shape (layouts as in ``gemm_ops``); problem i uses
    ``a + i sa``, ``b + i sb``, ``c + i sc`` (elements).  Operands are tensors or ``Ref``s
    (never raw addresses: what the launch will touch is checked against what they hold).
    ``bias`` (problem 0's bf16 vector, one group row apart) is added to the result;
    ``c2`` receives ``gelu(c)`` with the strides of ``c``."""
    _positive(M=M, N=N, K=K, batch=batch, lda=lda, ldb=ldb, ldc=ldc)
    if min(sa, sb, sc) < 0:
        raise DeviceError("gemm_batched: negative problem stride")
    (ar, ac), (br, bc) = {gemm_ops.NT: ((M, K), (N, K)), gemm_ops.NN: ((M, K), (K, N)),
                          gemm_ops.TN: ((K, M), (K, N))}[layout]
    n8 = (N + 7) // 8 * 8  # the epilogue stores whole 16-byte pieces
    if lda < ac or ldb < bc or ldc < n8 or lda % 8 or ldb % 8 or ldc % 8:
        raise DeviceError(
            f"gemm_batched: row pitches {lda}, {ldb}, {ldc} must be multiples of 8 and cover "
            f"{ac}, {bc}, {n8} columns")
    a_ptr, a_have = _span(a, "gemm_batched A")
    b_ptr, b_have = _span(b, "gemm_batched B")
    c_ptr, c_have = _span(c, "gemm_batched C")
    for what, have, need in (("A", a_have, _extent(ar, ac, lda, batch, sa)),
                             ("B", b_have, _extent(br, bc, ldb, batch, sb)),
                             ("C", c_have, _extent(M, n8, ldc, batch, sc))):
        if have < need:
            raise DeviceError(f"gemm_batched {what}: {have} elements behind the operand, "
                              f"the launch touches {need}")
    if a_ptr % 16 or b_ptr % 16 or c_ptr % 16:
        raise DeviceError("gemm_batched: operands must be 16-byte aligned")
    c2_ptr = None
    if c2 is not None:
        c2_ptr, c2_have = _span(c2, "gemm_batched C2")
        if c2_have < _extent(M, n8, ldc, batch, sc) or c2_ptr % 16:
            raise DeviceError("gemm_batched C2: too small or misaligned for the strides of C")
    bias_ptr, sbias = None, 0
    if bias is not None:
        _need_ref(bias, "gemm_batched bias", BF16, N, batch)
        bias_ptr, sbias = bias.ptr.value, bias.stride
    geo = gemm_ops.choose_geometry(M, N, K, layout, batch)
    splits = gemm_ops.choose_splits(M, N, K, geo, batch)
    if splits > 1:
        _need(slabs, "gemm_batched split-K buffer", F32, batch * splits * M * N)
    library.check(
        _fn("nf_gemm_bf16_batched")(layout, geo, a_ptr, b_ptr, c_ptr, c2_ptr, bias_ptr,
                                    _p(slabs) if splits > 1 else None, M, N, K, lda, ldb, ldc,
                                    splits, batch, sa, sb, sc, sbias, stream),
        "nf_gemm_bf16_batched",
    )


# -- patch gather / scatter ---------------------------------------------------------------

def im2col(x: torch.Tensor, col: torch.Tensor, n: int, h: int, w: int, c: int, ho: int, wo: int,
           k: int, stride: int, pad: int, kp: int) -> None:
    _positive(n=n, h=h, w=w, c=c, ho=ho, wo=wo, k=k, stride=stride, kp=kp)
    if kp < k * k * c or kp % 8 or pad < 0:
        raise DeviceError(f"im2col: padded patch length {kp} for {k}x{k}x{c}")
    if (h + 2 * pad - k) // stride + 1 != ho or (w + 2 * pad - k) // stride + 1 != wo:
        raise DeviceError("im2col: output size does not follow from input, kernel, stride, pad")
    _need(x, "im2col x", BF16, n * h * w * c)
    _need(col, "im2col patch matrix", BF16, n * ho * wo * kp)
    library.check(
        _fn("nf_im2col")(_p(x), _p(col), n, h, w, c, ho, wo, k, k, stride, pad, kp, _stream(x)),
        "im2col",
    )


def col2im(col: torch.Tensor, dx: torch.Tensor, n: int, h: int, w: int, c: int, ho: int, wo: int,
           k: int, stride: int, pad: int, kp: int) -> None:
    _positive(n=n, h=h, w=w, c=c, ho=ho, wo=wo, k=k, stride=stride, kp=kp)
    if kp < k * k * c or kp % 8 or pad < 0:
        raise DeviceError(f"col2im: padded patch length {kp} for {k}x{k}x{c}")
    if (h + 2 * pad - k) // stride + 1 != ho or (w + 2 * pad - k) // stride + 1 != wo:
        raise DeviceError("col2im: output size does not follow from input, kernel, stride, pad")
    _need(col, "col2im patch matrix", BF16, n * ho * wo * kp)
    _need(dx, "col2im dx", BF16, n * h * w * c)
    library.check(
        _fn("nf_col2im")(_p(col), _p(dx), n, h, w, c, ho, wo, k, k, stride, pad, kp, _stream(col)),
        "col2im",
    )


def col2im_join_blocks(b: int, h: int, w: int, c: int) -> int:
    """Rows of partial sums ``col2im_join`` writes per group."""
    return int(_fn("nf_col2im_join_blocks")(b, h, w, c))


def col2im_join(col: torch.Tensor, dx: torch.Tensor, add: torch.Tensor | None, mask: torch.Tensor,
                xpre: torch.Tensor, partial: torch.Tensor, G: int, b: int, h: int, w: int, c: int,
                ho: int, wo: int, k: int, stride: int, pad: int, kp: int) -> None:
    """``col2im`` for G groups of ``b`` images that also finishes the gradient for the
    BatchNorm whose ReLU output the convolution read: ``dx = (col2im + add)`` where
    ``mask > 0``, else 0, and ``partial`` ``[G, col2im_join_blocks, 2, c]`` receives the sums of
    dx and dx * xpre that ``bn_bwd_pre`` takes."""
    _positive(G=G, b=b, h=h, w=w, c=c, ho=ho, wo=wo, k=k, stride=stride, kp=kp)
    if kp < k * k * c or kp % 8 or pad < 0:
        raise DeviceError(f"col2im_join: padded patch length {kp} for {k}x{k}x{c}")
    if (h + 2 * pad - k) // stride + 1 != ho or (w + 2 * pad - k) // stride + 1 != wo:
        raise DeviceError("col2im_join: output size does not follow from input, kernel, stride, pad")
    if c % 8 or 256 % (c // 8):
        raise DeviceError(f"col2im_join: {c} channels (c / 8 must divide 256)")
    n_in = G * b * h * w * c
    _need(col, "col2im_join patch matrix", BF16, G * b * ho * wo * kp)
    _need(dx, "col2im_join dx", BF16, n_in)
    _need(add, "col2im_join add", BF16, n_in, optional=True)
    _need(mask, "col2im_join mask", BF16, n_in)
    _need(xpre, "col2im_join xpre", BF16, n_in)
    _need(partial, "col2im_join partial sums", F32, G * col2im_join_blocks(b, h, w, c) * 2 * c)
    for t, what in ((add, "add"), (mask, "mask"), (xpre, "xpre")):
        if t is not None and t.data_ptr() == dx.data_ptr():
            raise DeviceError(f"col2im_join: dx must not alias {what}")
    library.check(
        _fn("nf_col2im_join_g")(_p(col), _p(dx), _p(add), _p(mask), _p(xpre), _p(partial), G, b, h,
                                w, c, ho, wo, k, k, stride, pad, kp, _stream(col)),
        "col2im_join_g",
    )


def stem_patches(images: torch.Tensor, rows: torch.Tensor, col: torch.Tensor, h: int, w: int,
                 scale, shift) -> None:
    """Patch matrix ``[N * h * w, 32]`` of the normalised images ``images[rows]``
    (uint8 ``[*, 3, h, w]``): entry ``(ky * 3 + kx) * 3 + c`` of a row is
    ``images[.., c, y + ky - 1, x + kx - 1] * scale[c] + shift[c]`` (0 outside)."""
    if images.dtype != torch.uint8 or not images.is_contiguous() or rows.dtype != torch.int64:
        raise DeviceError("stem_patches: contiguous uint8 images and int64 rows are required")
    if images.dim() != 4 or tuple(images.shape[1:]) != (3, h, w) or rows.dim() != 1:
        raise DeviceError("stem_patches: images [*, 3, h, w] and a vector of rows are required")
    if rows.device != images.device or not rows.is_contiguous():
        raise DeviceError("stem_patches: the row indices must be a contiguous tensor on the images' device")
    _need(col, "stem_patches patch matrix", BF16, rows.numel() * h * w * 32)
    sc = (c_float * 3)(*scale)
    sh = (c_float * 3)(*shift)
    library.check(
        _fn("nf_stem_patches")(_p(images), _p(rows), _p(col), rows.numel(), h, w, sc, sh,
                               _stream(images)),
        "stem_patches",
    )


# -- classifier head and optimizer --------------------------------------------------------

def pool_fc_ce(x: torch.Tensor, w: Ref, bias: Ref, labels: torch.Tensor, pooled: torch.Tensor,
               dlogits: torch.Tensor, loss_rows: torch.Tensor, correct_rows: torch.Tensor,
               dx: torch.Tensor | None, G: int, B: int, pixels: int, C: int, classes: int,
               train: bool = True) -> None:
    _positive(G=G, B=B, pixels=pixels, C=C, classes=classes)
    _need(x, "pool_fc_ce x", BF16, G * B * pixels * C)
    _need_ref(w, "pool_fc_ce w", (BF16, F32), classes * C, G)
    _need_ref(bias, "pool_fc_ce bias", (BF16, F32), classes, G)
    _need(labels, "pool_fc_ce labels", (torch.int32, torch.int64), G * B)
    _need(pooled, "pool_fc_ce pooled", (BF16, F32), G * B * C)
    _need(dlogits, "pool_fc_ce d(logits)", (BF16, F32), G * B * classes)
    _need(loss_rows, "pool_fc_ce loss rows", F32, G * B)
    _need(correct_rows, "pool_fc_ce correct rows", F32, G * B)
    _need(dx, "pool_fc_ce dx", BF16, G * B * pixels * C, optional=True)
    library.check(
        _fn("nf_pool_fc_ce")(_p(x), w.ptr, bias.ptr, _p(labels), _p(pooled), _p(dlogits),
                             _p(loss_rows), _p(correct_rows), _p(dx), G, B, pixels, C, classes,
                             w.stride, bias.stride, int(train), _stream(x)),
        "pool_fc_ce",
    )


def fc_grads(pooled: torch.Tensor, dlogits: torch.Tensor, loss_rows: torch.Tensor,
             correct_rows: torch.Tensor, dw: Ref, db: Ref, acc: torch.Tensor | None, G: int,
             B: int, C: int, classes: int) -> None:
    _positive(G=G, B=B, C=C, classes=classes)
    _need(pooled, "fc_grads pooled", (BF16, F32), G * B * C)
    _need(dlogits, "fc_grads d(logits)", (BF16, F32), G * B * classes)
    _need(loss_rows, "fc_grads loss rows", F32, G * B)
    _need(correct_rows, "fc_grads correct rows", F32, G * B)
    _need_ref(dw, "fc_grads dw", (BF16, F32), classes * C, G)
    _need_ref(db, "fc_grads db", (BF16, F32), classes, G)
    _need(acc, "fc_grads accumulators", F32, G * 3, optional=True)
    library.check(
        _fn("nf_fc_grads")(_p(pooled), _p(dlogits), _p(loss_rows), _p(correct_rows), dw.ptr,
                           db.ptr, _p(acc), G, B, C, classes, dw.stride, _stream(pooled)),
        "fc_grads",
    )


def sgd_rows(p: torch.Tensor, g: torch.Tensor, shadow: torch.Tensor, n: int, lr: float,
             weight_decay: float) -> None:
    """Rows of ``p`` (float32), ``g`` and ``shadow`` (bf16), first ``n`` elements of
    each: ``p -= lr (g + wd p)``, ``shadow = bf16(p)``."""
    for t, what, dt in ((p, "p", F32), (g, "g", BF16), (shadow, "shadow", BF16)):
        if t.device.type not in _DEVICE_TYPES or t.dtype != dt or t.dim() != 2 or t.stride(1) != 1:
            raise DeviceError(f"sgd_rows {what}: a [rows, n] device matrix of {dt} is required")
        if t.shape[0] != p.shape[0] or t.shape[1] < n:
            raise DeviceError(f"sgd_rows {what}: shape {tuple(t.shape)} for {p.shape[0]} rows of {n}")
    if n < 0 or n % 4:
        raise DeviceError("sgd_rows: the row length must be a multiple of 4")
    library.check(
        _fn("nf_sgd_rows")(_p(p), _p(g), _p(shadow), p.shape[0], n, p.stride(0), g.stride(0),
                           shadow.stride(0), lr, weight_decay, _stream(p)),
        "sgd_rows",
    )
