"""gfx950 transformer kernels (csrc/hip/nn_ops.hip, attention.hip) against the
fp32 PyTorch implementations of the same ops (the CPU paths of
nanofed_amd.ops.nn_ops / attention)."""

import pytest
import torch
import torch.nn.functional as F

from nanofed_amd.ops import attention as attn_ops
from nanofed_amd.ops import nn_ops

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda:0"


def bf(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


@pytest.mark.parametrize("R,C", [(4096, 768), (37, 64), (130, 1600), (64, 2048)])
def test_layernorm(R, C):
    x, d, g, b, dy, res = (bf(R, C, seed=s) for s in range(1, 7)) if False else (
        bf(R, C, seed=1), bf(R, C, seed=2), bf(C, seed=3), bf(C, seed=4), bf(R, C, seed=5),
        bf(R, C, seed=6))
    outs = {}
    for dev in ("cpu", DEV):
        t = [v.to(dev) for v in (x, d, g, b, dy, res)]
        y, xo = torch.empty(R, C, dtype=BF, device=dev), torch.empty(R, C, dtype=BF, device=dev)
        mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
        nn_ops.ln_fwd(t[0], t[2], t[3], y, mean, rstd, delta=t[1], x_out=xo)
        y2 = torch.empty_like(y)
        m2, r2 = torch.empty_like(mean), torch.empty_like(rstd)
        nn_ops.ln_fwd(xo, t[2], t[3], y2, m2, r2)
        dx, dg, db = (torch.empty(R, C, dtype=BF, device=dev), torch.empty(C, dtype=BF, device=dev),
                      torch.empty(C, dtype=BF, device=dev))
        dxs = torch.empty(C, dtype=BF, device=dev)
        nn_ops.ln_bwd(t[4], xo, t[2], mean, rstd, dx, dg, db, dres=t[5], dxsum=dxs)
        dx2 = torch.empty_like(dx)
        nn_ops.ln_bwd(t[4], xo, t[2], mean, rstd, dx2, dg, db)
        outs[dev] = [v.float().cpu() for v in (xo, y, y2, mean, rstd, dx, dx2, dg, db, dxs)]
    names = ["x_out", "y", "y_nodelta", "mean", "rstd", "dx_res", "dx", "dgamma", "dbeta", "dxsum"]
    for name, a, b_ in zip(names, outs["cpu"], outs[DEV]):
        tol = dict(rtol=2e-2, atol=2e-2) if name not in ("dgamma", "dbeta", "dxsum") else dict(
            rtol=2e-2, atol=2e-2 * max(1.0, R**0.5 / 4))
        torch.testing.assert_close(b_, a, **tol, msg=lambda m, n=name: f"{n}: {m}")
    assert torch.equal(outs["cpu"][0], outs[DEV][0])  # residual add rounds identically


def test_gelu_colsum_sgd():
    R, N = 4096, 3072
    x, dy = bf(R, N, seed=1), bf(R, N, seed=2)
    xd, dyd = x.to(DEV), dy.to(DEV)
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    nn_ops.gelu_fwd(xd, y)
    nn_ops.gelu_bwd(xd, dyd, dx)
    leaf = x.float().requires_grad_(True)
    ref = F.gelu(leaf, approximate="tanh")
    ref.backward(dy.float())
    torch.testing.assert_close(y.float().cpu(), ref.detach(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(dx.float().cpu(), leaf.grad, rtol=2e-2, atol=2e-2)
    # fused with the bias gradient, in place (dx aliases dy), as the trainer uses it
    dbias = torch.empty(N, dtype=BF, device=DEV)
    dy2 = dyd.clone()
    nn_ops.gelu_bwd_colsum(xd, dy2, dy2, dbias)
    assert torch.equal(dy2, dx)
    torch.testing.assert_close(dbias.float(), dx.float().sum(0), rtol=1e-2, atol=0.5)
    nn_ops.gelu_bwd(xd, dyd, dyd)
    assert torch.equal(dyd, dx)
    for rows, cols, ld in ((4096, 768, 768), (4096, 2304, 2304), (100, 50, 64), (4092, 50257 - 1, 50304)):
        m = bf(rows, ld, seed=3).to(DEV)
        out = torch.empty(cols, dtype=BF, device=DEV)
        nn_ops.colsum(m[:, :cols], out)
        want = m[:, :cols].float().sum(0)
        torch.testing.assert_close(out.float(), want, rtol=1e-2, atol=0.5)
    n = 1 << 20
    p = torch.randn(n)
    g = bf(n, seed=4)
    pd, gd, sh = p.to(DEV), g.to(DEV), torch.empty(n, dtype=BF, device=DEV)
    nn_ops.sgd_step_bf16g(pd, gd, sh, 0.1, 0.01)
    want = p - 0.1 * (g.float() + 0.01 * p)
    torch.testing.assert_close(pd.cpu(), want, rtol=1e-6, atol=1e-6)
    assert torch.equal(sh.cpu(), pd.cpu().to(BF))


# GPT-2 vocabulary, small rows, one row piece, long rows
@pytest.mark.parametrize("R,V", [(512, 50257), (33, 1000), (8, 8), (16, 20000), (4, 60000)])
def test_cross_entropy(R, V):
    ld = (V + 63) // 64 * 64
    logits = torch.full((R, ld), 50.0, dtype=BF)
    logits[:, :V] = bf(R, V, scale=4.0, seed=1)
    tgt = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(2))
    ref_logits = logits.clone()
    loss_c, hit_c = torch.empty(R), torch.empty(R)
    nn_ops.cross_entropy_fwd_bwd(ref_logits, tgt, V, 1.0 / R, loss_c, hit_c)
    ld_ = logits.to(DEV)
    loss, hit = torch.empty(R, device=DEV), torch.empty(R, device=DEV)
    nn_ops.cross_entropy_fwd_bwd(ld_, tgt.to(DEV), V, 1.0 / R, loss, hit)
    torch.testing.assert_close(loss.cpu(), loss_c, rtol=1e-3, atol=1e-3)
    assert torch.equal(hit.cpu(), hit_c)
    torch.testing.assert_close(ld_.float().cpu(), ref_logits.float(), rtol=2e-2, atol=1e-5)
    assert torch.count_nonzero(ld_[:, V:]) == 0


def test_embedding():
    V, T, C, B = 5000, 128, 768, 4
    wte, wpe = bf(V, C, seed=1), bf(T, C, seed=2)
    toks = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(3))
    toks[0, :40] = 7  # a long run of one token
    x = nn_ops.embed_fwd(toks.reshape(-1).to(DEV), wte.to(DEV), wpe.to(DEV), T)
    xc = nn_ops.embed_fwd(toks.reshape(-1), wte, wpe, T)
    assert torch.equal(x.cpu(), xc)
    dx = bf(B * T, C, seed=4)
    base = bf(V, C, seed=5)
    dwte, dwpe = base.clone(), torch.zeros(T, C, dtype=BF)
    nn_ops.embed_bwd(toks, dx, dwte, dwpe, B, T)
    gw, gp = base.to(DEV), torch.zeros(T, C, dtype=BF, device=DEV)
    nn_ops.embed_bwd(toks.to(DEV), dx.to(DEV), gw, gp, B, T)
    torch.testing.assert_close(gw.float().cpu(), dwte.float(), rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(gp.float().cpu(), dwpe.float(), rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize("B,T,H", [(1, 128, 1), (2, 256, 3), (4, 1024, 12)])
def test_causal_attention_forward_and_backward(B, T, H):
    C = H * 64
    qkv = bf(B * T, 3 * C, seed=1)
    dout = bf(B * T, C, seed=2)
    leaf = qkv.to(DEV).float().requires_grad_(True)
    q, k, v = leaf.view(B, T, 3, H, 64).unbind(2)
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                         is_causal=True).transpose(1, 2).reshape(B * T, C)
    ref.backward(dout.to(DEV).float())
    scores = (q.transpose(1, 2) @ k.transpose(1, 2).transpose(-1, -2)).detach() / 8.0
    mask = torch.ones(T, T, dtype=torch.bool, device=DEV).tril()
    lse_ref = torch.logsumexp(scores.masked_fill(~mask, float("-inf")), -1)

    qd, dd = qkv.to(DEV), dout.to(DEV)
    out = torch.empty(B * T, C, dtype=BF, device=DEV)
    lse = torch.empty(B, H, T, device=DEV)
    attn_ops.causal_attention_fwd(qd, out, lse, B, T, H)
    torch.testing.assert_close(out.float(), ref.detach(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(lse, lse_ref, rtol=1e-3, atol=1e-3)
    dqkv = torch.full_like(qd, float("nan"))
    attn_ops.causal_attention_bwd(qd, out, lse, dd, dqkv,
                                  attn_ops.backward_scratch(B, H, T, qd.device), B, T, H)
    got, want = dqkv.float(), leaf.grad
    assert torch.isfinite(got).all()
    err = (got - want).norm() / want.norm()
    assert err < 2e-2, float(err)
    torch.testing.assert_close(got, want, rtol=5e-2, atol=3e-2)


def test_attention_rejects_unsupported_lengths():
    from nanofed_amd.core.exceptions import DeviceError

    with pytest.raises(DeviceError):
        attn_ops.causal_attention_fwd(bf(100, 192).to(DEV), torch.empty(100, 64, dtype=BF, device=DEV),
                                      torch.empty(100, device=DEV), 1, 100, 1)
