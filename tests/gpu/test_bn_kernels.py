"""Fused BatchNorm + residual + ReLU (csrc/hip/bn_ops.hip) against PyTorch's
fp32 batch_norm / add / relu with autograd."""

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from nanofed_amd.ops.bn_ops import batch_norm_act, fused_path_applies

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(64, 64, 32, 32), (8, 128, 16, 16), (5, 256, 7, 9), (64, 512, 4, 4)])
@pytest.mark.parametrize("relu,with_res", [(True, False), (True, True), (False, False), (False, True)])
def test_bn_act_forward_backward(shape, relu, with_res):
    torch.manual_seed(0)
    n, c, h, w = shape
    x = (torch.randn(shape) * 1.5 + 0.3).to(torch.bfloat16)
    res = torch.randn(shape).to(torch.bfloat16) if with_res else None
    dy = torch.randn(shape).to(torch.bfloat16)

    # oracle: fp32 PyTorch on the same (bf16-rounded) inputs
    bn_ref = nn.BatchNorm2d(c).to(DEV)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5)
        bn_ref.bias.uniform_(-0.5, 0.5)
    xr = x.to(DEV).float().requires_grad_(True)
    rr = res.to(DEV).float().requires_grad_(True) if with_res else None
    out = bn_ref(xr)
    if with_res:
        out = out + rr
    if relu:
        out = F.relu(out)
    out.backward(dy.to(DEV).float())

    bn = nn.BatchNorm2d(c).to(DEV)
    bn.load_state_dict({k: v.clone() for k, v in bn_ref.state_dict().items()})
    with torch.no_grad():
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0)
        bn.num_batches_tracked.zero_()
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rd = res.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
    assert fused_path_applies(xd, True, rd)
    y = batch_norm_act(bn, xd, residual=rd, relu=relu)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(dy.to(DEV).contiguous(memory_format=torch.channels_last))

    torch.testing.assert_close(y.float(), out.detach(), rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(bn.running_mean, bn_ref.running_mean, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(bn.running_var, bn_ref.running_var, rtol=1e-3, atol=1e-3)
    assert int(bn.num_batches_tracked) == 1

    def rel(a, b):
        return float((a.float() - b).norm() / (b.norm() + 1e-12))

    # y is rounded to bf16 before the ReLU mask is taken, so a few elements next to
    # zero flip: compare in norm
    assert rel(xd.grad, xr.grad) < 3e-2
    assert rel(bn.weight.grad, bn_ref.weight.grad) < 2e-2
    assert rel(bn.bias.grad, bn_ref.bias.grad) < 2e-2
    if with_res:
        assert rel(rd.grad, rr.grad) < 2e-2


def test_fallbacks_keep_pytorch_semantics():
    bn = nn.BatchNorm2d(64).to(DEV)
    x = torch.randn(4, 64, 8, 8, device=DEV)  # fp32, NCHW: not the fused path
    assert not fused_path_applies(x, True, None)
    want = F.relu(nn.BatchNorm2d(64).to(DEV)(x))
    torch.testing.assert_close(batch_norm_act(bn, x), want)
    bn.eval()
    xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert not fused_path_applies(xb, bn.training, None)
    assert batch_norm_act(bn, xb).shape == x.shape
    odd = torch.randn(2, 24, 4, 4, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert not fused_path_applies(odd, True, None)


@pytest.mark.parametrize("with_res,relu", [(True, True), (False, True), (False, False)])
def test_inference_mode_uses_the_running_statistics(with_res, relu):
    """Evaluation without gradients runs the apply kernel alone; the oracle is
    F.batch_norm in eval mode on the same inputs in fp32."""
    torch.manual_seed(11)
    c, shape = 128, (8, 128, 8, 8)
    bn = nn.BatchNorm2d(c).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.normal_(0.0, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
    bn.eval()
    x = (torch.randn(shape, device=DEV) * 1.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn(shape, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if with_res else None
    before = (bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked))
    with torch.no_grad():
        y = batch_norm_act(bn, x, residual=res, relu=relu)
        ref = F.batch_norm(x.float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        if with_res:
            ref = ref + res.float()
        if relu:
            ref = F.relu(ref)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=2e-2)
    assert torch.equal(bn.running_mean, before[0]) and torch.equal(bn.running_var, before[1])
    assert int(bn.num_batches_tracked) == before[2]
