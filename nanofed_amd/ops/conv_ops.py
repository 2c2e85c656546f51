"""3x3 / stride 1 / padding 1 convolution as a differentiable op on the
hand-written MFMA kernels of csrc/hip/conv_ops.hip (channels-last bf16
activations on the GPU).

``conv2d(conv, x)`` takes the ``nn.Conv2d`` module whose weight it uses, so
state-dict keys stay those of PyTorch.  Whenever the kernels do not apply (CPU,
fp32 or NCHW activations, other kernel sizes / strides, a bias, channel counts
that are not multiples of 64) it calls the module, i.e. the library
convolution, which is also the oracle of the kernel test.  The reference has
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


# Weight gradient: the hand-written kernel wins for the ResNet stages with images of
# 8x8 and larger (26 / 24 / 24 us against 29 / 26 / 26 for the autotuned library at
# batch 64), the library for the 4x4 stage (22 against 26 us).  "auto" picks by image
# size; NANOFED_CONV_WGRAD=native|library forces one.
WEIGHT_GRAD = os.environ.get("NANOFED_CONV_WGRAD", "auto")


def use_native_weight_grad(x: torch.Tensor) -> bool:
    if WEIGHT_GRAD == "auto":
        return x.shape[2] * x.shape[3] >= 64
    return WEIGHT_GRAD == "native"


def library_weight_grad(x: torch.Tensor, dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return torch.ops.aten.convolution_backward(
        dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]
    )[1]


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, with_sums):
        w = weight.detach().contiguous(memory_format=CL)
        ctx.save_for_backward(x, w)
        if not with_sums:
            return conv3x3_forward(x, w)
        y, sums = conv3x3_forward(x, w, with_sums=True)
        ctx.mark_non_differentiable(sums)
        return y, sums

    @staticmethod
    def backward(ctx, dy, *unused):
        x, w = ctx.saved_tensors
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=CL):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=CL)
        dx = conv3x3_input_grad(dy, w) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = (conv3x3_weight_grad(x, dy) if use_native_weight_grad(x)
                  else library_weight_grad(x, dy, w))
        return dx, dw, None


def conv2d(conv: nn.Conv2d, x: torch.Tensor, with_sums: bool = False):
    """``conv(x)``; on the hand-written kernels when they cover the case.
    ``with_sums=True`` returns ``(y, sums)``: ``sums`` are the per-channel partial
    sums of ``conv3x3_forward`` on the kernel path and ``None`` otherwise."""
    if native_path_applies(conv, x):
        return _Conv3x3.apply(x, conv.weight.to(torch.bfloat16), with_sums)
    y = conv(x)
    return (y, None) if with_sums else y
