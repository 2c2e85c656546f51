"""Patch gather / scatter semantics (the CPU implementations that serve as oracles of
the HIP kernels): a convolution is ``im2col(x) · Wᵀ`` with the weight in its
channels-last storage order, and ``col2im`` is the adjoint of ``im2col``."""

import pytest
import torch
import torch.nn.functional as F

from nanofed_amd.ops import conv_ops


@pytest.mark.parametrize("c,o,k,s,p,hw", [(8, 16, 3, 2, 1, 8), (16, 8, 1, 2, 0, 8), (3, 8, 3, 1, 1, 6),
                                          (8, 8, 3, 1, 1, 4)])
def test_convolution_is_patch_matrix_times_weight(c, o, k, s, p, hw):
    torch.manual_seed(0)
    x = torch.randn(2, c, hw, hw).contiguous(memory_format=torch.channels_last)
    w = torch.randn(o, c, k, k).contiguous(memory_format=torch.channels_last)
    col = conv_ops.im2col(x, k, k, s, p)
    kp = col.shape[1]
    assert kp % 8 == 0 and kp >= k * k * c
    w2d = conv_ops._weight2d(w, kp)
    ho = (hw + 2 * p - k) // s + 1
    y = (col @ w2d.t()).view(2, ho, ho, o).permute(0, 3, 1, 2)
    torch.testing.assert_close(y, F.conv2d(x, w, stride=s, padding=p), rtol=1e-5, atol=1e-5)
    assert float(col[:, k * k * c:].abs().sum()) == 0.0


@pytest.mark.parametrize("c,k,s,p,hw", [(8, 3, 2, 1, 8), (8, 1, 2, 0, 8), (8, 3, 1, 1, 4)])
def test_col2im_is_the_adjoint_of_im2col(c, k, s, p, hw):
    torch.manual_seed(1)
    x = torch.randn(2, c, hw, hw).contiguous(memory_format=torch.channels_last)
    col = conv_ops.im2col(x, k, k, s, p)
    g = torch.randn_like(col)
    back = conv_ops.col2im(g, tuple(x.shape), k, k, s, p)
    # <im2col(x), g> == <x, col2im(g)>
    torch.testing.assert_close((col * g).sum(), (x * back).sum(), rtol=1e-4, atol=1e-4)
