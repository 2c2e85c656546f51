"""Convolutions as differentiable ops on the hand-written kernels (channels-last
bf16 activations on the GPU):

* 3x3 / stride 1 / padding 1 with channel counts that are multiples of 64: the
  implicit-GEMM MFMA kernels of csrc/hip/conv_ops.hip;
* every other bias-free, ungrouped convolution (strided 3x3, 1x1 shortcuts, the
  3-channel stem): patch gather (csrc/hip/im2col_ops.hip) + the bf16 GEMM of
  csrc/hip/gemm_ops.hip - forward ``col · Wᵀ``, weight gradient ``dyᵀ · col``, data
  gradient ``col2im(dy · W)``; the weight's channels-last storage
  ``[Cout][KH][KW][Cin]`` is the GEMM operand as it is.

``conv2d(conv, x)`` takes the ``nn.Conv2d`` module whose weight it uses, so
state-dict keys stay those of PyTorch.  Whenever the kernels do not apply (CPU,
fp32 or NCHW activations outside autocast, a bias, groups, dilation) it calls the
module, i.e. the library convolution, which is also the oracle of the kernel tests.  The reference has
no convolution beyond its MNIST model (nanofed/models/mnist.py:17-19); this op
serves the ResNet-18 configuration of the benchmark suite.
"""

from __future__ import annotations

import os
from ctypes import c_int, c_long, c_void_p

import torch
import torch.nn as nn

from . import library

P, L, I = c_void_p, c_long, c_int
_SIGS = {
    "nf_conv3x3_supported": ([I, I, I, I, I], I),
    "nf_conv3x3_fwd": ([P, P, P, P, I, I, I, I, I, P], I),
    "nf_conv3x3_fwd_tiles": ([I, I, I, I], I),
    "nf_conv3x3_dgrad": ([P, P, P, I, I, I, I, I, P], I),
    "nf_conv3x3_flip_weights": ([P, P, I, I, P], I),
    "nf_conv3x3_wgrad_workspace": ([I, I, I, I, I], L),
    "nf_conv3x3_wgrad": ([P, P, P, P, I, I, I, I, I, P], I),
    "nf_conv3x3_set_wgrad_min_workgroups": ([I], None),
    "nf_im2col": ([P, P, I, I, I, I, I, I, I, I, I, I, I, P], I),
    "nf_col2im": ([P, P, I, I, I, I, I, I, I, I, I, I, I, P], I),
}
_fns: dict[str, object] = {}
CL = torch.channels_last


def _fn(name: str):
    fn = _fns.get(name)
    if fn is None:
        argtypes, restype = _SIGS[name]
        fn = library.declare(name, argtypes, restype)
        _fns[name] = fn
    return fn


def symbols() -> list[str]:
    return sorted(_SIGS)


def _p(t: torch.Tensor) -> c_void_p:
    return c_void_p(t.data_ptr())


def shape_supported(batch: int, height: int, width: int, c_in: int, c_out: int) -> bool:
    return bool(_fn("nf_conv3x3_supported")(batch, height, width, c_in, c_out))


def set_wgrad_min_workgroups(n: int) -> None:
    """Tuning knob of the weight gradient's pixel split (default 192)."""
    _fn("nf_conv3x3_set_wgrad_min_workgroups")(int(n))


def native_path_applies(conv: nn.Conv2d, x: torch.Tensor) -> bool:
    if not (x.device.type == "cuda" and x.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    if not x.is_contiguous(memory_format=CL):
        return False
    if (
        conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1)
        or conv.dilation != (1, 1) or conv.groups != 1 or conv.bias is not None
        or conv.padding_mode != "zeros"
    ):
        return False
    b, c, h, w = x.shape
    return c == conv.in_channels and shape_supported(b, h, w, c, conv.out_channels)


def conv3x3_forward(x: torch.Tensor, w: torch.Tensor, with_sums: bool = False):
    """x: [B, Cin, H, W], w: [Cout, Cin, 3, 3], both bf16 in channels-last storage.
    ``with_sums``: also return [tiles, 2, Cout] float32 per-channel sums of the
    output and its square over disjoint pixel tiles (for a following BatchNorm)."""
    b, c_in, h, wd = x.shape
    c_out = w.shape[0]
    y = torch.empty((b, c_out, h, wd), dtype=torch.bfloat16, device=x.device, memory_format=CL)
    sums = None
    if with_sums:
        tiles = int(_fn("nf_conv3x3_fwd_tiles")(b, h, wd, c_out))
        sums = torch.empty((tiles, 2, c_out), dtype=torch.float32, device=x.device)
    library.check(
        _fn("nf_conv3x3_fwd")(_p(x), _p(w), _p(y), _p(sums) if sums is not None else c_void_p(0),
                              b, h, wd, c_in, c_out, library.stream_ptr(x.device)),
        "conv3x3_fwd",
    )
    return (y, sums) if with_sums else y


def conv3x3_input_grad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Gradient with respect to the input: dy [B, Cout, H, W] against the forward
    weight w [Cout, Cin, 3, 3] as it is (the kernel reads it transposed, with the taps
    reversed)."""
    b, c_out, h, wd = dy.shape
    c_in = w.shape[1]
    dx = torch.empty((b, c_in, h, wd), dtype=torch.bfloat16, device=dy.device, memory_format=CL)
    library.check(
        _fn("nf_conv3x3_dgrad")(_p(dy), _p(w), _p(dx), b, h, wd, c_in, c_out,
                                library.stream_ptr(dy.device)),
        "conv3x3_dgrad",
    )
    return dx


def flipped_weights(w: torch.Tensor) -> torch.Tensor:
    """[Cin, Cout, 3, 3] weights with reversed taps: ``conv3x3_forward(dy, flipped)`` is the
    input gradient too (the path ``conv3x3_input_grad`` took before it read ``w``
    directly; kept as a cross-check of the two kernels)."""
    c_out, c_in = w.shape[0], w.shape[1]
    wt = torch.empty((c_in, c_out, 3, 3), dtype=torch.bfloat16, device=w.device, memory_format=CL)
    library.check(
        _fn("nf_conv3x3_flip_weights")(_p(w), _p(wt), c_in, c_out, library.stream_ptr(w.device)),
        "conv3x3_flip_weights",
    )
    return wt


def conv3x3_weight_grad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    b, c_in, h, wd = x.shape
    c_out = dy.shape[1]
    elems = int(_fn("nf_conv3x3_wgrad_workspace")(b, h, wd, c_in, c_out))
    if elems <= 0:
        raise RuntimeError("conv3x3 weight gradient: unsupported shape")
    ws = torch.empty(elems, dtype=torch.bfloat16, device=x.device)
    dw = torch.empty((c_out, c_in, 3, 3), dtype=torch.bfloat16, device=x.device, memory_format=CL)
    library.check(
        _fn("nf_conv3x3_wgrad")(_p(x), _p(dy), _p(dw), _p(ws), b, h, wd, c_in, c_out,
                                library.stream_ptr(x.device)),
        "conv3x3_wgrad",
    )
    return dw


# Weight gradient: the implicit-GEMM kernel wins for the ResNet stages with images of
# 8x8 and larger (26 / 24 / 24 us against 29 / 26 / 26 for the autotuned library at
# batch 64); for the 4x4 stage its pixel ranges are too short, and the gradient is taken
# as patch gather + GEMM (dyT . col, 24 us against 23 for the library).  "auto" picks by
# image size; NANOFED_CONV_WGRAD=native|library forces one.
WEIGHT_GRAD = os.environ.get("NANOFED_CONV_WGRAD", "auto")


def use_native_weight_grad(x: torch.Tensor) -> bool:
    if WEIGHT_GRAD == "auto":
        return x.shape[2] * x.shape[3] >= 64
    return WEIGHT_GRAD == "native"


def library_weight_grad(x: torch.Tensor, dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return torch.ops.aten.convolution_backward(
        dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]
    )[1]


def gemm_weight_grad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """3x3 / stride 1 weight gradient as ``dyT . im2col(x)``."""
    from . import gemm_ops

    c_out, c_in = dy.shape[1], x.shape[1]
    col = im2col(x, 3, 3, 1, 1)
    dw2d = torch.empty((c_out, col.shape[1]), dtype=torch.bfloat16, device=x.device)
    gemm_ops.linear_wgrad(dy.permute(0, 2, 3, 1).reshape(-1, c_out), col, dw2d)
    return dw2d[:, : 9 * c_in].reshape(c_out, 3, 3, c_in).permute(0, 3, 1, 2)


def gemm_ops_enabled(device: torch.device) -> bool:
    from . import gemm_ops

    return gemm_ops.enabled(device)


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, with_sums):
        w = weight.detach().contiguous(memory_format=CL)
        ctx.save_for_backward(x, w)
        # the gradient slot of the non-differentiable `sums` output stays None instead of
        # a zero tensor that autograd would have to fill in every backward pass
        ctx.set_materialize_grads(False)
        if not with_sums:
            return conv3x3_forward(x, w)
        y, sums = conv3x3_forward(x, w, with_sums=True)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, *unused):
        if dy is None:
            return None, None, None
        x, w = ctx.saved_tensors
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=CL):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=CL)
        dx = conv3x3_input_grad(dy, w) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if use_native_weight_grad(x):
                dw = conv3x3_weight_grad(x, dy)
            elif WEIGHT_GRAD != "library" and GEMM_PATH and gemm_ops_enabled(x.device):
                dw = gemm_weight_grad(x, dy)
            else:
                dw = library_weight_grad(x, dy, w)
        return dx, dw, None


# -- convolution as patch gather + GEMM ------------------------------------------------

# NANOFED_CONV_GEMM=0 sends these convolutions back to the library (A/B measurements)
GEMM_PATH = os.environ.get("NANOFED_CONV_GEMM", "1") != "0"


def _out_size(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


def im2col(x: torch.Tensor, kh: int, kw: int, stride: int, pad: int) -> torch.Tensor:
    """Patch matrix ``[B * Ho * Wo, Kp]`` of a channels-last bf16 ``x`` ``[B, C, H, W]``;
    column ``(ky * kw + kx) * C + c``, rows padded with zeros to a multiple of 8."""
    b, c, h, w = x.shape
    ho, wo = _out_size(h, kh, stride, pad), _out_size(w, kw, stride, pad)
    kp = -(-(kh * kw * c) // 8) * 8
    if x.device.type != "cuda":
        cols = torch.nn.functional.unfold(x.float(), (kh, kw), padding=pad, stride=stride)
        cols = cols.view(b, c, kh * kw, ho * wo).permute(0, 3, 2, 1).reshape(b * ho * wo, kh * kw * c)
        out = torch.zeros(b * ho * wo, kp, dtype=x.dtype)
        out[:, : kh * kw * c] = cols.to(x.dtype)
        return out
    col = torch.empty((b * ho * wo, kp), dtype=torch.bfloat16, device=x.device)
    library.check(
        _fn("nf_im2col")(_p(x), _p(col), b, h, w, c, ho, wo, kh, kw, stride, pad, kp,
                         library.stream_ptr(x.device)),
        "im2col",
    )
    return col


def col2im(col: torch.Tensor, shape: tuple[int, int, int, int], kh: int, kw: int, stride: int,
           pad: int) -> torch.Tensor:
    """Adjoint of ``im2col``: channels-last ``[B, C, H, W]`` whose every element is the
    sum of the patch entries that were copies of it."""
    b, c, h, w = shape
    ho, wo = _out_size(h, kh, stride, pad), _out_size(w, kw, stride, pad)
    kp = col.shape[1]
    if col.device.type != "cuda":
        cols = col[:, : kh * kw * c].float().view(b, ho * wo, kh * kw, c).permute(0, 3, 2, 1)
        dx = torch.nn.functional.fold(cols.reshape(b, c * kh * kw, ho * wo), (h, w), (kh, kw),
                                      padding=pad, stride=stride)
        return dx.to(col.dtype).contiguous(memory_format=CL)
    dx = torch.empty((b, c, h, w), dtype=torch.bfloat16, device=col.device, memory_format=CL)
    library.check(
        _fn("nf_col2im")(_p(col), _p(dx), b, h, w, c, ho, wo, kh, kw, stride, pad, kp,
                         library.stream_ptr(col.device)),
        "col2im",
    )
    return dx


def gemm_path_applies(conv: nn.Conv2d, x: torch.Tensor) -> bool:
    from . import gemm_ops

    if not (GEMM_PATH and x.device.type == "cuda" and x.dim() == 4 and gemm_ops.enabled(x.device)):
        return False
    if x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=CL):
        return False
    if (
        conv.dilation != (1, 1) or conv.groups != 1 or conv.bias is not None
        or conv.padding_mode != "zeros" or conv.stride[0] != conv.stride[1]
        or conv.padding[0] != conv.padding[1] or isinstance(conv.padding, str)
    ):
        return False
    # data gradients need whole 16-byte channel groups; the stem (3 channels) has none
    return x.shape[1] == conv.in_channels and (x.shape[1] % 8 == 0 or not x.requires_grad)


def _weight2d(w: torch.Tensor, kp: int) -> torch.Tensor:
    """``[Cout, Kp]`` GEMM operand of a channels-last ``[Cout, Cin, KH, KW]`` weight: a
    view of its storage when the row length needs no padding."""
    o, i, kh, kw = w.shape
    flat = w.permute(0, 2, 3, 1).reshape(o, kh * kw * i)
    if flat.shape[1] == kp and flat.stride(1) == 1 and flat.stride(0) == kp and flat.data_ptr() % 16 == 0:
        return flat
    out = torch.zeros((o, kp), dtype=w.dtype, device=w.device)
    out[:, : kh * kw * i] = flat
    return out


class _ConvGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        from . import gemm_ops

        w = weight.detach().contiguous(memory_format=CL)
        o, _, kh, kw = w.shape
        b, _, h, wd = x.shape
        ho, wo = _out_size(h, kh, stride, pad), _out_size(wd, kw, stride, pad)
        col = im2col(x, kh, kw, stride, pad)
        w2d = _weight2d(w, col.shape[1])
        y = torch.empty((b, o, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=CL)
        gemm_ops.linear_fwd(col, w2d, None, y.permute(0, 2, 3, 1).reshape(b * ho * wo, o))
        ctx.save_for_backward(col, w2d)
        ctx.geo = (tuple(x.shape), tuple(w.shape), stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import gemm_ops

        col, w2d = ctx.saved_tensors
        xshape, (o, i, kh, kw), stride, pad = ctx.geo
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=CL):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=CL)
        dy2d = dy.permute(0, 2, 3, 1).reshape(-1, o)
        dx = dw = None
        if ctx.needs_input_grad[1]:
            dw2d = torch.empty_like(w2d)
            gemm_ops.linear_wgrad(dy2d, col, dw2d)
            dw = dw2d[:, : kh * kw * i].reshape(o, kh, kw, i).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[0]:
            dcol = torch.empty_like(col)
            gemm_ops.linear_dgrad(dy2d, w2d, dcol)
            dx = col2im(dcol, xshape, kh, kw, stride, pad)
        return dx, dw, None, None


def conv2d(conv: nn.Conv2d, x: torch.Tensor, with_sums: bool = False):
    """``conv(x)``; on the hand-written kernels when they cover the case.
    ``with_sums=True`` returns ``(y, sums)``: ``sums`` are the per-channel partial
    sums of ``conv3x3_forward`` on the implicit-GEMM path and ``None`` otherwise."""
    if native_path_applies(conv, x):
        return _Conv3x3.apply(x, conv.weight.to(torch.bfloat16), with_sums)
    xg = x
    if (x.device.type == "cuda" and x.dtype == torch.float32 and x.dim() == 4
            and torch.is_autocast_enabled()):
        # the first layer of a network under autocast: the library would cast too
        xg = x.to(torch.bfloat16).contiguous(memory_format=CL)
    if gemm_path_applies(conv, xg):
        y = _ConvGemm.apply(xg, conv.weight.to(torch.bfloat16), conv.stride[0], conv.padding[0])
        return (y, None) if with_sums else y
    y = conv(x)
    return (y, None) if with_sums else y
