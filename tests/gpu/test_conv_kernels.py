"""3x3 convolution kernels (csrc/hip/conv_ops.hip) against the fp32 PyTorch
convolution on the same bf16-rounded operands."""

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CL = torch.channels_last
# (batch, height, width, c_in, c_out): the four ResNet-18 stages at batch 64,
# a non-square image, unequal channel counts, a small batch
SHAPES = [
    (64, 32, 32, 64, 64),
    (64, 16, 16, 128, 128),
    (64, 8, 8, 256, 256),
    (64, 4, 4, 512, 512),
    (4, 8, 16, 64, 128),
    (8, 16, 16, 128, 64),
    (2, 32, 32, 64, 64),
    # large launches: the forward / data-gradient kernels take 256-pixel tiles (four waves
    # of 64 pixels, no k-split) - one image band, one whole image, four images per tile
    (128, 32, 32, 64, 64),
    (256, 16, 16, 128, 128),
    (512, 8, 8, 256, 256),
    (512, 16, 16, 64, 128),
]


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


@pytest.fixture(scope="module")
def ops():
    from nanofed_amd.ops import conv_ops

    return conv_ops


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_and_gradients(ops, shape):
    b, h, w, c_in, c_out = shape
    assert ops.shape_supported(*shape)
    gen = torch.Generator(device="cuda").manual_seed(b * 1000 + h * 10 + c_in)
    x = torch.randn(b, c_in, h, w, device="cuda", generator=gen).to(torch.bfloat16)
    x = x.contiguous(memory_format=CL)
    wt = (torch.randn(c_out, c_in, 3, 3, device="cuda", generator=gen) / (3 * c_in**0.5))
    wt = wt.to(torch.bfloat16).contiguous(memory_format=CL)
    dy = torch.randn(b, c_out, h, w, device="cuda", generator=gen).to(torch.bfloat16)
    dy = dy.contiguous(memory_format=CL)

    xr = x.float().requires_grad_(True)
    wr = wt.float().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1)
    yr.backward(dy.float())

    y = ops.conv3x3_forward(x, wt)
    assert y.shape == yr.shape and y.is_contiguous(memory_format=CL)
    assert _rel(y, yr) < 6e-3  # bf16 rounding of the output
    dx = ops.conv3x3_input_grad(dy, wt)
    assert _rel(dx, xr.grad) < 6e-3
    # the same gradient through the forward kernel on an explicitly flipped weight copy
    dx2 = ops.conv3x3_forward(dy, ops.flipped_weights(wt))
    assert _rel(dx2, xr.grad) < 6e-3
    dw = ops.conv3x3_weight_grad(x, dy)
    assert dw.shape == wr.grad.shape
    assert _rel(dw, wr.grad) < 6e-3


def test_large_launches_take_the_256_pixel_tile(ops):
    """The tile is chosen from the number of workgroups it would make; the partial column
    sums a following BatchNorm reads come one row per tile."""
    tiles = lambda b, h, w, c: int(ops._fn("nf_conv3x3_fwd_tiles")(b, h, w, c))  # noqa: E731
    assert tiles(128, 32, 32, 64) == 128 * 1024 // 256
    assert tiles(64, 32, 32, 64) == 64 * 1024 // 128       # too few tiles of 256 pixels
    assert tiles(512, 8, 8, 256) == 512 * 64 // 256
    assert tiles(512, 4, 4, 512) == 512 * 16 // 128        # the patch of 16 images does not fit


def test_weight_split_does_not_change_the_result_much(ops):
    shape = (64, 16, 16, 128, 128)
    gen = torch.Generator(device="cuda").manual_seed(7)
    b, h, w, c_in, c_out = shape
    x = torch.randn(b, c_in, h, w, device="cuda", generator=gen).to(torch.bfloat16)
    x = x.contiguous(memory_format=CL)
    dy = torch.randn(b, c_out, h, w, device="cuda", generator=gen).to(torch.bfloat16)
    dy = dy.contiguous(memory_format=CL)
    try:
        ops.set_wgrad_min_workgroups(1)      # one pixel range
        one = ops.conv3x3_weight_grad(x, dy)
        ops.set_wgrad_min_workgroups(1024)   # as many ranges as possible
        many = ops.conv3x3_weight_grad(x, dy)
    finally:
        ops.set_wgrad_min_workgroups(192)
    assert _rel(many, one) < 6e-3
    # the same split twice is bit-identical (fixed summation order)
    again = ops.conv3x3_weight_grad(x, dy)
    assert torch.equal(again, ops.conv3x3_weight_grad(x, dy))


def test_module_wrapper_matches_library_convolution(ops):
    torch.manual_seed(0)
    conv = nn.Conv2d(64, 64, 3, 1, 1, bias=False).cuda()
    x = torch.randn(16, 64, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    x1 = x.clone().requires_grad_(True)
    assert ops.native_path_applies(conv, x1)
    y1 = ops.conv2d(conv, x1)
    y1.float().square().mean().backward()
    g_native, gx_native = conv.weight.grad.clone(), x1.grad.clone()
    conv.weight.grad = None
    x2 = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = conv(x2)
    y2.float().square().mean().backward()
    assert _rel(y1, y2) < 1e-2
    assert _rel(g_native, conv.weight.grad) < 2e-2
    assert _rel(gx_native, x2.grad) < 2e-2


def test_unsupported_cases_use_the_module(ops):
    conv = nn.Conv2d(64, 64, 3, 2, 1, bias=False).cuda()
    x = torch.randn(4, 64, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    assert not ops.native_path_applies(conv, x)
    conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False).cuda()
    x3 = torch.randn(4, 3, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    assert not ops.native_path_applies(conv1, x3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert ops.conv2d(conv, x).shape == (4, 64, 8, 8)


@pytest.mark.parametrize("shape", SHAPES[:5])
def test_channel_sums_of_the_output(ops, shape):
    b, h, w, c_in, c_out = shape
    gen = torch.Generator(device="cuda").manual_seed(99 + c_in)
    x = torch.randn(b, c_in, h, w, device="cuda", generator=gen).to(torch.bfloat16)
    x = x.contiguous(memory_format=CL)
    wt = (torch.randn(c_out, c_in, 3, 3, device="cuda", generator=gen) / (3 * c_in**0.5))
    wt = wt.to(torch.bfloat16).contiguous(memory_format=CL)
    y_plain = ops.conv3x3_forward(x, wt)
    y, sums = ops.conv3x3_forward(x, wt, with_sums=True)
    assert torch.equal(y, y_plain)
    assert sums.shape[1:] == (2, c_out) and sums.dtype == torch.float32
    yf = y.float()
    ref1 = yf.sum(dim=(0, 2, 3))
    ref2 = (yf * yf).sum(dim=(0, 2, 3))
    torch.testing.assert_close(sums[:, 0].sum(0), ref1, rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(sums[:, 1].sum(0), ref2, rtol=1e-4, atol=1e-2)


def test_batchnorm_takes_the_convolution_sums(ops):
    from nanofed_amd.ops.bn_ops import batch_norm_act

    torch.manual_seed(4)
    conv = nn.Conv2d(128, 128, 3, 1, 1, bias=False).cuda()
    bns = [nn.BatchNorm2d(128).cuda().train() for _ in range(2)]
    x = torch.randn(64, 128, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    res = torch.randn_like(x)
    y, sums = ops.conv2d(conv, x, with_sums=True)
    assert sums is not None
    a = batch_norm_act(bns[0], y, residual=res, partials=sums)
    b = batch_norm_act(bns[1], y, residual=res)
    torch.testing.assert_close(a.float(), b.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(bns[0].running_mean, bns[1].running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bns[0].running_var, bns[1].running_var, rtol=1e-4, atol=1e-5)
    # gradients flow through both ops
    a.float().square().mean().backward()
    assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all()
