"""Convolutions on the patch-gather + GEMM path (strided 3x3, 1x1 shortcut, 3-channel
stem) against the fp32 library convolution: forward, data gradient, weight gradient;
the HIP gather / scatter kernels against their PyTorch implementations."""

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from nanofed_amd.ops import conv_ops

pytestmark = pytest.mark.gpu
CL = torch.channels_last

# (Cin, Cout, kernel, stride, pad, image) - the ResNet-18 layers that are not 3x3 / stride 1
SHAPES = [
    (64, 128, 3, 2, 1, 32), (128, 256, 3, 2, 1, 16), (256, 512, 3, 2, 1, 8),
    (64, 128, 1, 2, 0, 32), (128, 256, 1, 2, 0, 16), (256, 512, 1, 2, 0, 8),
    (3, 64, 3, 1, 1, 32), (40, 24, 5, 3, 2, 17),
]


@pytest.mark.parametrize("c,k,s,p,hw", [(64, 3, 2, 1, 16), (128, 1, 2, 0, 8), (3, 3, 1, 1, 32), (40, 5, 3, 2, 17)])
def test_gather_and_scatter_kernels_match_their_pytorch_forms(c, k, s, p, hw):
    torch.manual_seed(0)
    x = torch.randn(4, c, hw, hw).to(torch.bfloat16).contiguous(memory_format=CL)
    col = conv_ops.im2col(x.cuda(), k, k, s, p)
    want = conv_ops.im2col(x, k, k, s, p)
    assert torch.equal(col.cpu(), want)
    if c % 8 == 0:
        g = torch.randn_like(want)
        got = conv_ops.col2im(g.cuda(), tuple(x.shape), k, k, s, p)
        ref = conv_ops.col2im(g.float(), tuple(x.shape), k, k, s, p)
        torch.testing.assert_close(got.float().cpu(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("cin,cout,k,s,p,hw", SHAPES)
def test_convolution_matches_the_fp32_library(cin, cout, k, s, p, hw):
    torch.manual_seed(cin + cout + k)
    B = 16
    conv = nn.Conv2d(cin, cout, k, s, p, bias=False).cuda().to(memory_format=CL)
    wb = conv.weight.detach().to(torch.bfloat16)
    x = torch.randn(B, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    needs_dx = cin % 8 == 0
    x.requires_grad_(needs_dx)
    conv.weight.data = wb.contiguous(memory_format=CL)
    assert conv_ops.gemm_path_applies(conv, x) and not conv_ops.native_path_applies(conv, x)
    y = conv_ops.conv2d(conv, x)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=CL)
    dy = torch.randn_like(y)
    y.backward(dy)
    xf = x.detach().float().requires_grad_(True)
    wf = wb.float().requires_grad_(True)
    yf = F.conv2d(xf, wf, stride=s, padding=p)
    yf.backward(dy.float())
    scale = float(yf.abs().max())
    torch.testing.assert_close(y.float(), yf, rtol=2e-2, atol=1e-2 * scale)
    gw = conv.weight.grad.float()
    torch.testing.assert_close(gw, wf.grad, rtol=2e-2, atol=1e-2 * float(wf.grad.abs().max()))
    if needs_dx:
        torch.testing.assert_close(x.grad.float(), xf.grad, rtol=2e-2,
                                   atol=1.5e-2 * float(xf.grad.abs().max()))


def test_resnet18_step_uses_no_library_convolution(monkeypatch):
    """Forward + backward of the ResNet-18 under autocast: every convolution goes through
    the hand-written paths (the library convolution op is never dispatched)."""
    from torch.utils._python_dispatch import TorchDispatchMode

    from nanofed_amd.models import build_model

    class Log(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.ops = []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            self.ops.append(str(func))
            return func(*args, **(kwargs or {}))

    torch.manual_seed(0)
    model = build_model("resnet18").cuda().to(memory_format=CL).train()
    for prm in model.parameters():
        if prm.dim() == 4:
            prm.data = prm.data.to(torch.bfloat16).contiguous(memory_format=CL)
    x = torch.randn(64, 3, 32, 32, device="cuda").contiguous(memory_format=CL)
    y = torch.randint(0, 10, (64,), device="cuda")
    with Log() as log, torch.autocast("cuda", dtype=torch.bfloat16):
        loss = F.cross_entropy(model(x), y)
        loss.backward()
    convs = [op for op in log.ops if "convolution" in op]
    assert convs == [], convs
    assert torch.isfinite(loss)
    assert all(p.grad is not None and torch.isfinite(p.grad.float()).all() for p in model.parameters())
