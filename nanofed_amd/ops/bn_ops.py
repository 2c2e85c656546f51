"""BatchNorm + residual add + ReLU as one differentiable op
(csrc/hip/bn_ops.hip) for channels-last bf16 activations on the GPU.

``batch_norm_act`` takes the ``nn.BatchNorm2d`` module whose parameters and
running statistics it uses, so state-dict keys and the update rule of the
buffers stay those of PyTorch.  Whenever the fused kernels do not apply (CPU,
evaluation mode, fp32 or NCHW activations, unusual channel counts) it runs
``F.batch_norm`` + add + ``F.relu``, which is also the oracle of the kernel test.
The reference has no normalisation layer; this serves the ResNet-18
configuration of the benchmark suite.
"""

from __future__ import annotations

from ctypes import c_float, c_int, c_long, c_void_p

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import library

P, L, I, FL = c_void_p, c_long, c_int, c_float
_SIGS = {
    "nf_bn_supported": ([I], I),
    "nf_bn_workspace_floats": ([I], I),
    "nf_bn_fwd": ([P, P, P, P, P, P, P, P, P, L, I, I, FL, FL, I, P], I),
    "nf_bn_bwd": ([P, P, P, P, P, P, P, P, P, P, L, I, I, P], I),
    "nf_bn_apply": ([P, P, P, P, P, P, L, I, I, P], I),
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


def channels_supported(channels: int) -> bool:
    return bool(_fn("nf_bn_supported")(channels))


def fused_path_applies(x: torch.Tensor, training: bool, residual: torch.Tensor | None) -> bool:
    """Layout / dtype conditions of the kernels (training: also called with the module's
    mode; the inference path passes True and checks the mode itself)."""
    if not (training and x.device.type == "cuda" and x.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    if not x.is_contiguous(memory_format=torch.channels_last):
        return False
    if residual is not None and (
        residual.dtype != torch.bfloat16 or residual.shape != x.shape
        or not residual.is_contiguous(memory_format=torch.channels_last)
    ):
        return False
    return bool(_fn("nf_bn_supported")(x.shape[1]))


class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, relu, momentum, eps,
                partials):
        n, c, h, w = x.shape
        rows = n * h * w
        dev = x.device
        y = torch.empty_like(x)  # channels-last like x
        stats = torch.empty(2 * c, dtype=torch.float32, device=dev)
        if partials is not None:  # [blocks, 2, C] sums of x and x^2 from x's producer
            if partials.dtype != torch.float32 or partials.shape[1:] != (2, c) or (
                    not partials.is_contiguous()):
                raise ValueError("partial sums must be contiguous float32 [blocks, 2, C]")
            ws, pre = partials, partials.shape[0]
        else:
            ws = torch.empty(int(_fn("nf_bn_workspace_floats")(c)), dtype=torch.float32,
                             device=dev)
            pre = 0
        gamma, beta = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        library.check(
            _fn("nf_bn_fwd")(_p(x), _p(residual), _p(y), _p(gamma), _p(beta), _p(running_mean),
                             _p(running_var), _p(stats), _p(ws), rows, c, int(relu), momentum,
                             eps, pre, library.stream_ptr(dev)),
            "bn_fwd",
        )
        ctx.save_for_backward(x, y, stats, gamma)
        ctx.relu, ctx.has_res = relu, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, stats, gamma = ctx.saved_tensors
        n, c, h, w = x.shape
        rows = n * h * w
        dev = x.device
        if dy.dtype != torch.bfloat16 or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        ws = torch.empty(int(_fn("nf_bn_workspace_floats")(c)) + 2 * c, dtype=torch.float32,
                         device=dev)
        library.check(
            _fn("nf_bn_bwd")(_p(dy), _p(y), _p(x), _p(stats), _p(gamma), _p(dx), _p(dres),
                             _p(dgamma), _p(dbeta), _p(ws), rows, c, int(ctx.relu),
                             library.stream_ptr(dev)),
            "bn_bwd",
        )
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None


def batch_norm_act(bn: nn.BatchNorm2d, x: torch.Tensor, residual: torch.Tensor | None = None,
                   relu: bool = True, partials: torch.Tensor | None = None) -> torch.Tensor:
    """``relu(batch_norm(x) + residual)`` with ``bn``'s parameters and buffers.

    ``partials`` ([blocks, 2, C] float32 sums of x and x^2 over disjoint pixel
    ranges, as ``conv_ops.conv2d(..., with_sums=True)`` returns them) spares the
    statistics pass over ``x``; it is ignored on the fallback path.

    A trainer that counts optimisation steps itself sets ``bn.defer_batch_count``
    and adds the count to ``num_batches_tracked`` once per round, which saves one
    tiny kernel per layer and step."""
    if bn.affine and bn.track_running_stats and bn.momentum is not None and fused_path_applies(
        x, bn.training, residual
    ):
        if not getattr(bn, "defer_batch_count", False):
            with torch.no_grad():
                bn.num_batches_tracked.add_(1)
        return _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   residual, relu, float(bn.momentum), float(bn.eps), partials)
    if (not bn.training and bn.affine and bn.track_running_stats and not torch.is_grad_enabled()
            and fused_path_applies(x, True, residual)):
        return _inference(bn, x, residual, relu)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out) if relu else out


def _inference(bn: nn.BatchNorm2d, x: torch.Tensor, residual: torch.Tensor | None,
               relu: bool) -> torch.Tensor:
    """Evaluation mode without gradients: the apply kernel alone, on the running statistics."""
    n, c, h, w = x.shape
    stats = torch.stack((bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)))
    y = torch.empty_like(x)
    library.check(
        _fn("nf_bn_apply")(_p(x), _p(residual), _p(y), _p(bn.weight.detach().float().contiguous()),
                           _p(bn.bias.detach().float().contiguous()), _p(stats.contiguous()),
                           n * h * w, c, int(relu), library.stream_ptr(x.device)),
        "bn_apply",
    )
    return y
