"""CPU implementations of the transformer ops against PyTorch autograd: these
are the oracles the GPU kernel tests compare with, so they are checked first."""

import math

import pytest
import torch
import torch.nn.functional as F

from nanofed_amd.ops import attention as attn_ops
from nanofed_amd.ops import nn_ops

BF = torch.bfloat16


def bf(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def test_layernorm_forward_and_backward_match_autograd():
    R, C = 12, 64
    x, d, g, b, dy = bf(R, C, seed=1), bf(R, C, seed=2), bf(C, seed=3), bf(C, seed=4), bf(R, C, seed=5)
    y, xo = torch.empty(R, C, dtype=BF), torch.empty(R, C, dtype=BF)
    mean, rstd = torch.empty(R), torch.empty(R)
    nn_ops.ln_fwd(x, g, b, y, mean, rstd, delta=d, x_out=xo)
    assert torch.equal(xo, (x.float() + d.float()).to(BF))
    leaf = xo.float().requires_grad_(True)
    gl, bl = g.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = F.layer_norm(leaf, (C,), gl, bl)
    torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=2e-2)
    ref.backward(dy.float())
    dx, dg, db = torch.empty(R, C, dtype=BF), torch.empty(C, dtype=BF), torch.empty(C, dtype=BF)
    res = bf(R, C, seed=6)
    nn_ops.ln_bwd(dy, xo, g, mean, rstd, dx, dg, db, dres=res)
    torch.testing.assert_close(dx.float(), leaf.grad + res.float(), rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(dg.float(), gl.grad, rtol=3e-2, atol=5e-2)
    torch.testing.assert_close(db.float(), bl.grad, rtol=3e-2, atol=5e-2)


def test_gelu_and_colsum_and_sgd():
    x, dy = bf(8, 32, seed=1), bf(8, 32, seed=2)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    nn_ops.gelu_fwd(x, y)
    leaf = x.float().requires_grad_(True)
    ref = F.gelu(leaf, approximate="tanh")
    torch.testing.assert_close(y.float(), ref, rtol=1e-2, atol=1e-2)
    ref.backward(dy.float())
    nn_ops.gelu_bwd(x, dy, dx)
    torch.testing.assert_close(dx.float(), leaf.grad, rtol=2e-2, atol=2e-2)
    out = torch.empty(32, dtype=BF)
    nn_ops.colsum(x, out)
    torch.testing.assert_close(out.float(), x.float().sum(0), rtol=1e-2, atol=2e-2)
    p = torch.randn(64)
    g = bf(64, seed=3)
    shadow = torch.empty(64, dtype=BF)
    want = p - 0.1 * (g.float() + 0.01 * p)
    nn_ops.sgd_step_bf16g(p, g, shadow, 0.1, 0.01)
    torch.testing.assert_close(p, want)
    assert torch.equal(shadow, want.to(BF))


def test_cross_entropy_loss_accuracy_and_gradient():
    R, V, ld = 10, 37, 64
    logits = torch.zeros(R, ld, dtype=BF)
    logits[:, :V] = bf(R, V, scale=3.0, seed=1)
    logits[:, V:] = 100.0  # padding must be ignored
    tgt = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(2))
    leaf = logits[:, :V].float().requires_grad_(True)
    ref = F.cross_entropy(leaf, tgt, reduction="none")
    ref.sum().backward()
    loss, hit = torch.empty(R), torch.empty(R)
    nn_ops.cross_entropy_fwd_bwd(logits, tgt, V, 1.0 / R, loss, hit)
    torch.testing.assert_close(loss, ref.detach(), rtol=1e-4, atol=1e-4)
    assert torch.equal(hit, (leaf.argmax(1) == tgt).float())
    torch.testing.assert_close(logits[:, :V].float(), leaf.grad / R, rtol=2e-2, atol=1e-3)
    assert torch.count_nonzero(logits[:, V:]) == 0


def test_embedding_forward_and_backward():
    V, T, C, B = 50, 8, 16, 3
    wte, wpe = bf(V, C, seed=1), bf(T, C, seed=2)
    toks = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(3))
    x = nn_ops.embed_fwd(toks.reshape(-1), wte, wpe, T)
    want = wte[toks].float() + wpe.float()[None]
    torch.testing.assert_close(x.view(B, T, C).float(), want, rtol=1e-2, atol=1e-2)
    dx = bf(B * T, C, seed=4)
    dwte, dwpe = bf(V, C, seed=5), torch.zeros(T, C, dtype=BF)
    base = dwte.float().clone()
    nn_ops.embed_bwd(toks, dx, dwte, dwpe, B, T)
    ref = base.index_add(0, toks.reshape(-1), dx.float())
    torch.testing.assert_close(dwte.float(), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dwpe.float(), dx.view(B, T, C).float().sum(0), rtol=2e-2, atol=2e-2)


def test_attention_cpu_path_matches_sdpa_and_autograd():
    B, T, H = 2, 16, 2
    C = H * 64
    qkv = bf(B * T, 3 * C, seed=1)
    out = torch.empty(B * T, C, dtype=BF)
    lse = torch.empty(B, H, T)
    attn_ops.causal_attention_fwd(qkv, out, lse, B, T, H)
    leaf = qkv.float().requires_grad_(True)
    q, k, v = leaf.view(B, T, 3, H, 64).unbind(2)
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                         is_causal=True).transpose(1, 2).reshape(B * T, C)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)
    scores = (q.transpose(1, 2) @ k.transpose(1, 2).transpose(-1, -2)).detach() / 8.0
    assert lse[0, 0, 0].item() == pytest.approx(scores[0, 0, 0, 0].item(), rel=1e-5)
    dout = bf(B * T, C, seed=2)
    ref.backward(dout.float())
    dqkv = torch.empty_like(qkv)
    attn_ops.causal_attention_bwd(qkv, out, lse, dout, dqkv,
                                  attn_ops.backward_scratch(B, H, T, qkv.device), B, T, H)
    torch.testing.assert_close(dqkv.float(), leaf.grad, rtol=2e-2, atol=2e-2)


def test_operand_checks():
    from nanofed_amd.core.exceptions import DeviceError

    with pytest.raises(DeviceError):
        nn_ops.embed_fwd(torch.zeros(4, dtype=torch.int32), bf(4, 8), bf(4, 8), 4)
    with pytest.raises(DeviceError):
        attn_ops.causal_attention_fwd(bf(8, 100), bf(8, 32), torch.empty(8), 1, 8, 1)
    assert math.isclose(attn_ops.HEAD_DIM**-0.5, 0.125)
