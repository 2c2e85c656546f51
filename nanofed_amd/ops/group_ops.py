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
    "nf_conv3x3_wgrad_workspace_g": ([I, I, I, I, I, I], L),
    "nf_conv3x3_wgrad_g": ([P, P, P, P, I, I, I, I, I, I, L, P], I),
    "nf_bn_fwd_g": ([P, P, P, P, P, P, P, P, P, I, L, I, I, FL, FL, I, L, P], I),
    "nf_bn_bwd_g": ([P, P, P, P, P, P, P, P, P, P, P, I, L, I, I, L, I, L, P], I),
    "nf_bn_workspace_floats": ([I], I),
    "nf_gemm_bf16_batched": ([I, I, P, P, P, P, P, P, I, I, I, L, L, L, I, I, L, L, L, L, P], I),
    "nf_im2col": ([P, P, I, I, I, I, I, I, I, I, I, I, I, P], I),
    "nf_col2im": ([P, P, I, I, I, I, I, I, I, I, I, I, I, P], I),
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
    library.check(
        _fn("nf_conv3x3_fwd_g")(_p(x), w.ptr, _p(y), _p(colsum), G, B, H, W, c_in, c_out,
                                w.stride, _stream(x)),
        "conv3x3_fwd_g",
    )


def conv3x3_dgrad(dy: torch.Tensor, w: Ref, dx: torch.Tensor, G: int, B: int, H: int, W: int,
                  c_in: int, c_out: int) -> None:
    library.check(
        _fn("nf_conv3x3_dgrad_g")(_p(dy), w.ptr, _p(dx), G, B, H, W, c_in, c_out, w.stride,
                                  _stream(dy)),
        "conv3x3_dgrad_g",
    )


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: Ref, workspace: torch.Tensor, G: int,
                  B: int, H: int, W: int, c_in: int, c_out: int) -> None:
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
    library.check(
        _fn("nf_bn_bwd_g")(_p(dy), _p(dy2), _p(y if relu else None), _p(x), _p(stats), gamma.ptr,
                           _p(dx), _p(dres), dgamma.ptr, dbeta.ptr, _p(workspace), G, rows, C,
                           int(relu), gamma.stride, int(out_bf16), dgamma.stride, _stream(x)),
        "bn_bwd_g",
    )


# -- batched GEMM -------------------------------------------------------------------------

def gemm_slab_floats(M: int, N: int, K: int, layout: int, batch: int) -> int:
    geo = gemm_ops.choose_geometry(M, N, K, layout, batch)
    splits = gemm_ops.choose_splits(M, N, K, geo, batch)
    return batch * splits * M * N if splits > 1 else 0


def gemm_batched(layout: int, a: c_void_p, b: c_void_p, c: c_void_p, M: int, N: int, K: int,
                 lda: int, ldb: int, ldc: int, batch: int, sa: int, sb: int, sc: int,
                 slabs: torch.Tensor | None, stream: c_void_p, bias: int | None = None,
                 sbias: int = 0, c2: int | None = None) -> None:
    """``batch`` GEMMs of one shape (layouts as in ``gemm_ops``); problem i uses
    ``a + i sa``, ``b + i sb``, ``c + i sc`` (elements).  ``bias`` (address of problem
    0's bf16 vector, ``sbias`` elements apart) is added to the result; ``c2`` receives
    ``gelu(c)`` with the strides of ``c``."""
    geo = gemm_ops.choose_geometry(M, N, K, layout, batch)
    splits = gemm_ops.choose_splits(M, N, K, geo, batch)
    if splits > 1 and (slabs is None or slabs.numel() < batch * splits * M * N):
        raise DeviceError("gemm_batched: the split-K buffer is too small")
    library.check(
        _fn("nf_gemm_bf16_batched")(layout, geo, a, b, c, c2, bias,
                                    _p(slabs) if splits > 1 else None, M, N, K, lda, ldb, ldc,
                                    splits, batch, sa, sb, sc, sbias, stream),
        "nf_gemm_bf16_batched",
    )


# -- patch gather / scatter ---------------------------------------------------------------

def im2col(x: torch.Tensor, col: torch.Tensor, n: int, h: int, w: int, c: int, ho: int, wo: int,
           k: int, stride: int, pad: int, kp: int) -> None:
    library.check(
        _fn("nf_im2col")(_p(x), _p(col), n, h, w, c, ho, wo, k, k, stride, pad, kp, _stream(x)),
        "im2col",
    )


def col2im(col: torch.Tensor, dx: torch.Tensor, n: int, h: int, w: int, c: int, ho: int, wo: int,
           k: int, stride: int, pad: int, kp: int) -> None:
    library.check(
        _fn("nf_col2im")(_p(col), _p(dx), n, h, w, c, ho, wo, k, k, stride, pad, kp, _stream(col)),
        "col2im",
    )


def stem_patches(images: torch.Tensor, rows: torch.Tensor, col: torch.Tensor, h: int, w: int,
                 scale, shift) -> None:
    """Patch matrix ``[N * h * w, 32]`` of the normalised images ``images[rows]``
    (uint8 ``[*, 3, h, w]``): entry ``(ky * 3 + kx) * 3 + c`` of a row is
    ``images[.., c, y + ky - 1, x + kx - 1] * scale[c] + shift[c]`` (0 outside)."""
    if images.dtype != torch.uint8 or not images.is_contiguous() or rows.dtype != torch.int64:
        raise DeviceError("stem_patches: contiguous uint8 images and int64 rows are required")
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
    library.check(
        _fn("nf_pool_fc_ce")(_p(x), w.ptr, bias.ptr, _p(labels), _p(pooled), _p(dlogits),
                             _p(loss_rows), _p(correct_rows), _p(dx), G, B, pixels, C, classes,
                             w.stride, bias.stride, int(train), _stream(x)),
        "pool_fc_ce",
    )


def fc_grads(pooled: torch.Tensor, dlogits: torch.Tensor, loss_rows: torch.Tensor,
             correct_rows: torch.Tensor, dw: Ref, db: Ref, acc: torch.Tensor | None, G: int,
             B: int, C: int, classes: int) -> None:
    library.check(
        _fn("nf_fc_grads")(_p(pooled), _p(dlogits), _p(loss_rows), _p(correct_rows), dw.ptr,
                           db.ptr, _p(acc), G, B, C, classes, dw.stride, _stream(pooled)),
        "fc_grads",
    )


def sgd_rows(p: torch.Tensor, g: torch.Tensor, shadow: torch.Tensor, n: int, lr: float,
             weight_decay: float) -> None:
    """Rows of ``p`` (float32), ``g`` and ``shadow`` (bf16), first ``n`` elements of
    each: ``p -= lr (g + wd p)``, ``shadow = bf16(p)``."""
    library.check(
        _fn("nf_sgd_rows")(_p(p), _p(g), _p(shadow), p.shape[0], n, p.stride(0), g.stride(0),
                           shadow.stride(0), lr, weight_decay, _stream(p)),
        "sgd_rows",
    )
