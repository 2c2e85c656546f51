"""Client-grouped kernels and the ResNet-18 trainer built on them, against fp32
PyTorch references of the same ops (per group) and against the per-client trainer."""

import pytest
import torch
import torch.nn.functional as F

from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.ops import gemm_ops
from nanofed_amd.ops import group_ops as go
from nanofed_amd.ops.group_ops import Ref
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF = torch.bfloat16


def _cos(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.float().flatten(), b.float().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp(min=1e-30))


def _rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


# Shapes of the four ResNet-18 stages; B * H * W is a multiple of 128 in each.
STAGES = [(2, 8, 32, 64, 64), (3, 8, 16, 128, 128), (2, 16, 8, 256, 256), (2, 32, 4, 512, 512),
          (2, 8, 16, 64, 128)]


@pytest.mark.parametrize("xorder", ["0", "1"])
@pytest.mark.parametrize("G,B,H,cin,cout", STAGES)
def test_grouped_conv3x3_against_the_library_per_group(G, B, H, cin, cout, xorder, monkeypatch):
    # both workgroup orders of the forward / data-gradient launch (pixel tiles fastest, or
    # the tiles that share a weight slice kept on one XCD)
    monkeypatch.setenv("NANOFED_CONV_XORDER", xorder)
    torch.manual_seed(G * 1000 + H)
    assert go.conv3x3_supported(G, B, H, H, cin, cout)
    row = cout * 9 * cin + 64  # rows of a pretend arena: a gap after the tensor
    wbuf = (torch.randn(G, row, device=DEV) * 0.05).to(BF)
    x = torch.randn(G * B, cin, H, H, device=DEV).to(BF).contiguous(memory_format=torch.channels_last)
    w = Ref(wbuf, 0)
    y = torch.empty(G * B, cout, H, H, device=DEV, dtype=BF).contiguous(memory_format=torch.channels_last)
    tiles = go.conv3x3_tiles(G, B, H, H, cout)
    colsum = torch.empty(G * tiles, 2, cout, device=DEV)
    go.conv3x3_fwd(x, w, y, colsum, G, B, H, H, cin, cout)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    go.conv3x3_dgrad(dy, w, dx, G, B, H, H, cin, cout)
    gbuf = torch.zeros(G, row, device=DEV, dtype=BF)
    ws = torch.empty(go.conv3x3_wgrad_workspace(G, B, H, H, cin, cout), device=DEV, dtype=BF)
    go.conv3x3_wgrad(x, dy, Ref(gbuf, 0), ws, G, B, H, H, cin, cout)
    for g in range(G):
        wg = w.view(g, (cout, 3, 3, cin)).permute(0, 3, 1, 2).float()
        xg = x[g * B : (g + 1) * B].float().requires_grad_(True)
        wg.requires_grad_(True)
        ref = F.conv2d(xg, wg, padding=1)
        ref.backward(dy[g * B : (g + 1) * B].float())
        assert _rel(y[g * B : (g + 1) * B], ref) < 6e-3
        assert _rel(dx[g * B : (g + 1) * B], xg.grad) < 6e-3
        got = Ref(gbuf, 0).view(g, (cout, 3, 3, cin)).permute(0, 3, 1, 2)
        assert _rel(got, wg.grad) < 1.5e-2
        yb = y[g * B : (g + 1) * B].float()
        sums = colsum[g * tiles : (g + 1) * tiles].sum(0)
        torch.testing.assert_close(sums[0], yb.sum(dim=(0, 2, 3)), rtol=2e-3, atol=0.5)
        torch.testing.assert_close(sums[1], (yb * yb).sum(dim=(0, 2, 3)), rtol=2e-3, atol=0.5)
    assert float(gbuf[:, cout * 9 * cin :].abs().max()) == 0.0  # nothing outside the slot


@pytest.mark.parametrize("C,rows,relu,res,two", [(64, 2048, True, True, True),
                                                  (128, 512, True, False, False),
                                                  (512, 128, False, False, True),
                                                  (256, 1000, True, True, False)])
def test_grouped_batchnorm_against_autograd_per_group(C, rows, relu, res, two):
    torch.manual_seed(C + rows)
    G = 3
    prow = 4 * C + 64
    params = torch.zeros(G, prow, device=DEV)
    params[:, :C] = torch.rand(G, C, device=DEV) + 0.5           # gamma
    params[:, C : 2 * C] = torch.randn(G, C, device=DEV) * 0.1   # beta
    params[:, 3 * C : 4 * C] = 1.0                                # running_var
    gamma, beta = Ref(params, 0), Ref(params, C)
    rmean, rvar = Ref(params, 2 * C), Ref(params, 3 * C)
    x = (torch.randn(G * rows, C, device=DEV) * 2 + 0.3).to(BF)
    r = torch.randn(G * rows, C, device=DEV).to(BF) if res else None
    y = torch.empty_like(x)
    stats = torch.empty(G, 2, C, device=DEV)
    ws = torch.empty(go.bn_workspace_floats(G, C), device=DEV)
    go.bn_fwd(x, r, y, gamma, beta, rmean, rvar, stats, ws, G, rows, C, relu, 0.1, 1e-5)
    dy = torch.randn_like(x)
    dy2 = torch.randn_like(x) if two else None
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if res else None
    grads = torch.zeros(G, prow, device=DEV, dtype=BF)
    go.bn_bwd(dy, dy2, y, x, stats, gamma, dx, dres, Ref(grads, 0), Ref(grads, C), ws, G, rows, C,
              relu)
    for g in range(G):
        sl = slice(g * rows, (g + 1) * rows)
        xg = x[sl].float().requires_grad_(True)
        ga = params[g, :C].clone().requires_grad_(True)
        be = params[g, C : 2 * C].clone().requires_grad_(True)
        rg = r[sl].float().requires_grad_(True) if res else None
        out = F.batch_norm(xg, None, None, ga, be, training=True, eps=1e-5)
        if res:
            out = out + rg
        if relu:
            out = F.relu(out)
        d = dy[sl].float() + (dy2[sl].float() if two else 0)
        # the kernel masks with the bf16 output it stored; do the same in the reference
        if relu:
            d = d * (y[sl].float() > 0)
            out = F.batch_norm(xg, None, None, ga, be, training=True, eps=1e-5)
            if res:
                out = out + rg
        out.backward(d)
        ref_y = F.batch_norm(x[sl].float(), None, None, params[g, :C], params[g, C : 2 * C],
                             training=True, eps=1e-5)
        if res:
            ref_y = ref_y + r[sl].float()
        if relu:
            ref_y = F.relu(ref_y)
        assert _rel(y[sl], ref_y) < 6e-3
        assert _rel(dx[sl], xg.grad) < 1e-2
        if res:
            assert _rel(dres[sl], rg.grad) < 6e-3
        assert _rel(grads[g, :C], ga.grad) < 1e-2
        assert _rel(grads[g, C : 2 * C], be.grad) < 1e-2
        mean = x[sl].float().mean(0)
        var = x[sl].float().var(0, unbiased=True)
        torch.testing.assert_close(params[g, 2 * C : 3 * C], 0.1 * mean, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(params[g, 3 * C : 4 * C], 0.9 + 0.1 * var, rtol=1e-3, atol=1e-4)


# every wave arrangement of the data-gradient kernel: 256-pixel tiles (6 and 7 patch pieces),
# 128-pixel tiles, 64-pixel tiles
FUSED = [(8, 64, 32, 64, 64, True), (8, 64, 8, 256, 256, False), (2, 8, 16, 128, 128, True),
         (2, 16, 8, 256, 256, False), (2, 32, 4, 512, 512, True), (3, 8, 16, 128, 64, True)]


@pytest.mark.parametrize("G,B,H,cin,cout,two", FUSED)
def test_data_gradient_with_the_batchnorm_backward_epilogue(G, B, H, cin, cout, two):
    """conv3x3_dgrad(fuse=...) + bn_bwd_pre against the separate kernels they replace
    (conv3x3_dgrad, then bn_bwd with dy2 / ReLU mask / d(residual)) and against fp32
    autograd of BatchNorm on the same incoming gradient."""
    torch.manual_seed(G * 100 + H + cin)
    N, C, rows = G * B, cin, B * H * H
    wrow = cout * 9 * cin + 64
    w = Ref((torch.randn(G, wrow, device=DEV) * 0.05).to(BF), 0)
    dy = torch.randn(N, H, H, cout, device=DEV).to(BF)
    # the BatchNorm that produced the convolution's input: x -> y (ReLU output)
    prow = 2 * C + 64
    params = torch.zeros(G, prow, device=DEV)
    params[:, :C] = torch.rand(G, C, device=DEV) + 0.5
    params[:, C : 2 * C] = torch.randn(G, C, device=DEV) * 0.1
    gamma, beta = Ref(params, 0), Ref(params, C)
    x = (torch.randn(N, H, H, C, device=DEV) * 2 + 0.3).to(BF)
    y = torch.empty_like(x)
    stats = torch.empty(G, 2, C, device=DEV)
    ws = torch.empty(go.bn_workspace_floats(G, C), device=DEV)
    go.bn_fwd(x, None, y, gamma, beta, None, None, stats, ws, G, rows, C, True, 0.1, 1e-5)
    other = torch.randn_like(x) if two else None

    # separate kernels
    dx_plain = torch.empty_like(x)
    go.conv3x3_dgrad(dy, w, dx_plain, G, B, H, H, cin, cout)
    dxa = torch.empty_like(x)
    dres = torch.empty_like(x)
    ga = torch.zeros(G, prow, device=DEV, dtype=BF)
    go.bn_bwd(dx_plain, other, y, x, stats, gamma, dxa, dres, Ref(ga, 0), Ref(ga, C), ws, G, rows,
              C, True)

    # fused
    tiles = go.conv3x3_tiles(G, B, H, H, cin)
    pre = torch.full((G * tiles, 2, C), float("nan"), device=DEV)
    gz = torch.empty_like(x)
    go.conv3x3_dgrad(dy, w, gz, G, B, H, H, cin, cout,
                     fuse=go.BnBwdFuse(mask=y, xpre=x, add=other, colsum=pre))
    dxb = torch.empty_like(x)
    gb = torch.zeros(G, prow, device=DEV, dtype=BF)
    go.bn_bwd_pre(gz, x, stats, gamma, dxb, Ref(gb, 0), Ref(gb, C), pre, tiles, ws, G, rows, C)

    assert torch.isfinite(pre).all()
    # the masked, joined gradient has the roundings of the separate path: the same bits
    assert torch.equal(gz, dres)
    # ... and so has the variant that recomputes the mask from the BatchNorm's input
    # (this BatchNorm has no residual input) instead of reading its output
    pre2 = torch.full_like(pre, float("nan"))
    gz2 = torch.empty_like(x)
    go.conv3x3_dgrad(dy, w, gz2, G, B, H, H, cin, cout,
                     fuse=go.BnBwdFuse(mask=None, xpre=x, add=other, colsum=pre2, stats=stats,
                                       gamma=gamma, beta=beta))
    assert torch.equal(gz2, gz) and torch.equal(pre2, pre)
    assert bool(((gz == 0) | (y > 0)).all())
    assert _rel(dxb, dxa) < 1e-2
    assert _rel(gb[:, : 2 * C], ga[:, : 2 * C]) < 1e-2
    for g in range(G):
        sl = slice(g * B, (g + 1) * B)
        # partial sums are sums over the group's tiles of gz and gz * x
        gf, xf = gz[sl].float().view(-1, C), x[sl].float().view(-1, C)
        sums = pre[g * tiles : (g + 1) * tiles].sum(0)
        torch.testing.assert_close(sums[0], gf.sum(0), rtol=2e-3, atol=0.5)
        torch.testing.assert_close(sums[1], (gf * xf).sum(0), rtol=2e-3, atol=0.5)
        # fp32 autograd of the BatchNorm for the gradient gz
        xg = xf.clone().requires_grad_(True)
        gm = params[g, :C].clone().requires_grad_(True)
        be = params[g, C : 2 * C].clone().requires_grad_(True)
        F.batch_norm(xg, None, None, gm, be, training=True, eps=1e-5).backward(gf)
        assert _rel(dxb[sl].view(-1, C), xg.grad) < 1e-2
        assert _rel(gb[g, :C], gm.grad) < 1e-2
        assert _rel(gb[g, C : 2 * C], be.grad) < 1e-2


@pytest.mark.parametrize("G,B,H,C,k,two", [(3, 8, 32, 64, 1, True), (2, 8, 16, 128, 3, True),
                                           (2, 16, 8, 256, 1, False)])
def test_scatter_with_the_batchnorm_backward_join(G, B, H, C, k, two):
    """col2im_join + bn_bwd_pre against col2im followed by bn_bwd (second branch, ReLU mask,
    d(residual)): the joined gradient has the same bits, the BatchNorm results agree."""
    torch.manual_seed(G + H + C + k)
    N, Ho, pad, rows = G * B, H // 2, k // 2, B * H * H
    kp = k * k * C
    col = torch.randn(N * Ho * Ho, kp, device=DEV).to(BF)
    prow = 2 * C + 64
    params = torch.zeros(G, prow, device=DEV)
    params[:, :C] = torch.rand(G, C, device=DEV) + 0.5
    params[:, C : 2 * C] = torch.randn(G, C, device=DEV) * 0.1
    gamma, beta = Ref(params, 0), Ref(params, C)
    x = (torch.randn(N, H, H, C, device=DEV) * 2 + 0.3).to(BF)
    res = torch.randn_like(x)
    y = torch.empty_like(x)
    stats = torch.empty(G, 2, C, device=DEV)
    ws = torch.empty(go.bn_workspace_floats(G, C), device=DEV)
    go.bn_fwd(x, res, y, gamma, beta, None, None, stats, ws, G, rows, C, True, 0.1, 1e-5)
    other = torch.randn_like(x) if two else None

    plain = torch.empty_like(x)
    go.col2im(col, plain, N, H, H, C, Ho, Ho, k, 2, pad, kp)
    dxa, dres = torch.empty_like(x), torch.empty_like(x)
    ga = torch.zeros(G, prow, device=DEV, dtype=BF)
    go.bn_bwd(plain, other, y, x, stats, gamma, dxa, dres, Ref(ga, 0), Ref(ga, C), ws, G, rows, C,
              True)

    nb = go.col2im_join_blocks(B, H, H, C)
    pre = torch.full((G * nb, 2, C), float("nan"), device=DEV)
    gz = torch.empty_like(x)
    go.col2im_join(col, gz, other, y, x, pre, G, B, H, H, C, Ho, Ho, k, 2, pad, kp)
    dxb = torch.empty_like(x)
    gb = torch.zeros(G, prow, device=DEV, dtype=BF)
    go.bn_bwd_pre(gz, x, stats, gamma, dxb, Ref(gb, 0), Ref(gb, C), pre, nb, ws, G, rows, C)
    assert torch.equal(gz, dres)
    assert torch.isfinite(pre).all()
    for g in range(G):
        gf = gz[g * B : (g + 1) * B].float().view(-1, C)
        xf = x[g * B : (g + 1) * B].float().view(-1, C)
        sums = pre[g * nb : (g + 1) * nb].sum(0)
        torch.testing.assert_close(sums[0], gf.sum(0), rtol=2e-3, atol=0.5)
        torch.testing.assert_close(sums[1], (gf * xf).sum(0), rtol=2e-3, atol=0.5)
    assert _rel(dxb, dxa) < 1e-2
    assert _rel(gb[:, : 2 * C], ga[:, : 2 * C]) < 1e-2


def test_fused_batchnorm_backward_trains_the_same_models(monkeypatch):
    """The trainer with the BatchNorm backward folded into the data gradients against the
    trainer with the separate passes: same gradients up to the order in which the
    per-channel sums are added, same round."""
    fused = _engine("native", graph=False, clients=2, samples=64, batch=32)
    assert fused.trainer._fuse_bn_bwd
    monkeypatch.setenv("NANOFED_RESNET_FUSE_BN_BWD", "0")
    plain = _engine("native", graph=False, clients=2, samples=64, batch=32)
    assert not plain.trainer._fuse_bn_bwd
    rows = torch.cat([torch.arange(32, device=DEV) + int(fused.trainer.shards.offsets[k])
                      for k in range(2)])
    grads = []
    for eng in (fused, plain):
        eng.trainer.pool.broadcast_from(eng.global_arena.data)
        loss, gb = eng.trainer.loss_and_grads(rows)
        grads.append((loss.clone(), gb.clone()))
    assert torch.equal(grads[0][0], grads[1][0])  # the forward pass is the same code
    lay = fused.trainer.layout
    for seg in lay.segments:
        if not seg.trainable:
            continue
        a = lay.view_of(grads[0][1][0], seg)
        b = lay.view_of(grads[1][1][0], seg)
        assert _cos(a, b) > 0.995, (seg.name, _cos(a, b))
    ra, rb = fused.run(1), plain.run(1)
    assert ra[0].agg_metrics["loss"] == pytest.approx(rb[0].agg_metrics["loss"], rel=2e-2)
    assert _rel(fused.global_arena.data, plain.global_arena.data) < 0.02


@pytest.mark.parametrize("layout,M,N,K", [(gemm_ops.NT, 1024, 128, 576), (gemm_ops.NT, 4096, 64, 32),
                                           (gemm_ops.NN, 512, 1152, 256), (gemm_ops.TN, 64, 32, 8192),
                                           (gemm_ops.TN, 512, 2304, 1024), (gemm_ops.TN, 128, 64, 4096)])
def test_batched_gemm_equals_a_loop_of_single_gemms(layout, M, N, K):
    torch.manual_seed(M + N + K)
    batch = 3
    pad = 64
    if layout == gemm_ops.NT:
        a_shape, b_shape = (M, K), (N, K)
    elif layout == gemm_ops.NN:
        a_shape, b_shape = (M, K), (K, N)
    else:
        a_shape, b_shape = (K, M), (K, N)
    a = torch.randn(batch, a_shape[0] * a_shape[1] + pad, device=DEV).to(BF)
    b = torch.randn(batch, b_shape[0] * b_shape[1] + pad, device=DEV).to(BF)
    c = torch.zeros(batch, M * N + pad, device=DEV, dtype=BF)
    slabs = torch.empty(max(go.gemm_slab_floats(M, N, K, layout, batch), 1), device=DEV)
    go.gemm_batched(layout, a, b, c, M, N, K, a_shape[1],
                    b_shape[1], N, batch, a.stride(0), b.stride(0), c.stride(0), slabs,
                    go._stream(c))
    for i in range(batch):
        am = a[i, : a_shape[0] * a_shape[1]].view(a_shape).float()
        bm = b[i, : b_shape[0] * b_shape[1]].view(b_shape).float()
        if layout == gemm_ops.NT:
            ref = am @ bm.t()
        elif layout == gemm_ops.NN:
            ref = am @ bm
        else:
            ref = am.t() @ bm
        got = c[i, : M * N].view(M, N)
        assert _rel(got, ref) < 5e-3
    assert float(c[:, M * N :].abs().max()) == 0.0


def test_stem_patches_equal_unfold_of_the_normalised_image():
    torch.manual_seed(1)
    images = torch.randint(0, 256, (40, 3, 32, 32), dtype=torch.uint8, device=DEV)
    rows = torch.randperm(40, device=DEV)[:16]
    mean, std = (0.49, 0.48, 0.45), (0.25, 0.24, 0.26)
    scale = [1 / (255 * s) for s in std]
    shift = [-m / s for m, s in zip(mean, std)]
    col = torch.empty(16 * 1024, 32, device=DEV, dtype=BF)
    go.stem_patches(images, rows, col, 32, 32, scale, shift)
    x = images[rows].float() / 255
    x = (x - torch.tensor(mean, device=DEV).view(1, 3, 1, 1)) / torch.tensor(std, device=DEV).view(1, 3, 1, 1)
    ref = F.unfold(x.to(BF).float(), 3, padding=1)            # [16, c * 9 + tap, 1024]
    ref = ref.view(16, 3, 9, 1024).permute(0, 3, 2, 1).reshape(16 * 1024, 27)
    torch.testing.assert_close(col[:, :27].float(), ref, rtol=1e-2, atol=1e-2)
    assert float(col[:, 27:].abs().max()) == 0.0


def test_head_against_autograd():
    torch.manual_seed(2)
    G, B, P, C, K = 3, 16, 16, 512, 10
    row = K * C + 64
    wbuf = (torch.randn(G, row, device=DEV) * 0.05).to(BF)
    bbuf = torch.randn(G, 64, device=DEV) * 0.1
    x = torch.randn(G * B, P, C, device=DEV).to(BF)
    labels = torch.randint(0, K, (G * B,), device=DEV)
    pooled = torch.empty(G * B, C, device=DEV, dtype=BF)
    dlog = torch.empty(G * B, 16, device=DEV)
    loss = torch.empty(G * B, device=DEV)
    hit = torch.empty(G * B, device=DEV)
    dx = torch.empty_like(x)
    go.pool_fc_ce(x, Ref(wbuf, 0), Ref(bbuf, 0), labels, pooled, dlog, loss, hit, dx, G, B, P, C, K)
    grads = torch.zeros(G, row + 64, device=DEV, dtype=BF)
    acc = torch.zeros(G, 3, device=DEV)
    go.fc_grads(pooled, dlog, loss, hit, Ref(grads, 0), Ref(grads, row), acc, G, B, C, K)
    for g in range(G):
        sl = slice(g * B, (g + 1) * B)
        xg = x[sl].float().requires_grad_(True)
        w = wbuf[g, : K * C].view(K, C).float().requires_grad_(True)
        b = bbuf[g, :K].clone().requires_grad_(True)
        logits = xg.mean(1) @ w.t() + b
        l = F.cross_entropy(logits, labels[sl])
        l.backward()
        assert float(loss[sl].mean()) == pytest.approx(float(l.detach()), rel=2e-2)
        assert _rel(dx[sl], xg.grad) < 3e-2
        assert _rel(grads[g, : K * C].view(K, C), w.grad) < 3e-2
        assert _rel(grads[g, row : row + K], b.grad) < 3e-2
        assert float(acc[g, 0]) == pytest.approx(float(loss[sl].mean()), rel=1e-5)
        assert float(acc[g, 2]) == 1.0
        pred = logits.argmax(1)
        agree = float((pred == labels[sl]).float().mean())
        assert abs(float(acc[g, 1]) - agree) <= 2.0 / B  # bf16 logits may flip a near tie


def test_sgd_rows():
    torch.manual_seed(3)
    p = torch.randn(3, 1024 + 64, device=DEV)
    g = torch.randn(3, 1024, device=DEV).to(BF)
    sh = torch.zeros(3, 1024, device=DEV, dtype=BF)
    ref = p.clone()
    ref[:, :1024] -= 0.1 * (g.float() + 0.01 * ref[:, :1024])
    go.sgd_rows(p, g, sh, 1024, 0.1, 0.01)
    torch.testing.assert_close(p, ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(sh, ref[:, :1024].to(BF))


def _engine(backend: str, graph: bool = True, clients: int = 2, samples=64, batch: int = 32,
            rounds_seed: int = 5, **kw):
    torch.cuda.set_device(0)
    cfg = EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=clients,
                       samples_per_client=samples, batch_size=batch, local_epochs=1,
                       eval_samples=64, learning_rate=0.05, seed=rounds_seed, backend=backend,
                       graph_capture=graph, **kw)
    return FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))


def test_auto_backend_picks_the_grouped_trainer():
    from nanofed_amd.engine.resnet_native import ResNetNativeTrainer

    eng = _engine("auto")
    assert isinstance(eng.trainer, ResNetNativeTrainer)


def test_any_cifar_shaped_image_set_gets_the_grouped_trainer(tmp_path):
    """The data decide, not the dataset's name: CIFAR-10 binary batches on disk (fixture
    files in the format) and the hard synthetic set both train on the grouped kernels."""
    import numpy as np

    from nanofed_amd.data import write_cifar10_batch
    from nanofed_amd.data.cifar import names_of
    from nanofed_amd.engine.resnet_native import ResNetNativeTrainer

    rng = np.random.default_rng(0)
    folder = tmp_path / "cifar-10-batches-bin"
    folder.mkdir()
    for name in names_of(True) + names_of(False):
        y = rng.integers(0, 10, size=64)
        x = (rng.integers(0, 96, size=(64, 3, 32, 32)) + 16 * y[:, None, None, None]).astype(np.uint8)
        write_cifar10_batch(folder / name, x, y)
    torch.cuda.set_device(0)
    for kw in (dict(dataset="cifar10", data_dir=str(tmp_path)), dict(dataset="synthetic_cifar10_hard")):
        cfg = EngineConfig(model="resnet18", num_clients=2, samples_per_client=64, batch_size=32,
                           local_epochs=1, eval_samples=64, learning_rate=0.05, seed=5, **kw)
        eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
        assert isinstance(eng.trainer, ResNetNativeTrainer), kw
        reps = eng.run(2)
        assert all(r.status == "COMPLETED" for r in reps)
        assert torch.isfinite(eng.global_arena.data).all()
        assert 0.0 < reps[-1].agg_metrics["loss"] < 4.0


def test_gradients_of_a_batch_against_fp32_autograd_on_the_same_weights():
    from nanofed_amd.models import build_model

    eng = _engine("native", graph=False, clients=2, samples=64, batch=32)
    tr = eng.trainer
    tr.pool.broadcast_from(eng.global_arena.data)
    # different weights per client, so that a mixed-up group would show
    tr.pool.data[1, : tr.layout.trainable_numel] *= 1.05
    rows = torch.cat([torch.arange(32, device=DEV) + int(tr.shards.offsets[k]) for k in range(2)])
    loss, gb = tr.loss_and_grads(rows)
    for k in range(2):
        # plain PyTorch, fp32, NCHW, on the bf16-rounded weights the kernels read
        ref = build_model("resnet18").to(DEV).float().train()
        state = {n: t.to(DEV) for n, t in tr.pool[k].state_dict().items()}
        for n, t in state.items():
            if t.dim() >= 2:
                state[n] = t.to(BF).float()
        ref.load_state_dict(state)
        x, y = tr.shards.batch(rows[k * 32 : (k + 1) * 32])
        l = F.cross_entropy(ref(x.to(BF).float()), y)
        params = dict(ref.named_parameters())
        g32 = dict(zip(params, torch.autograd.grad(l, list(params.values()))))
        # ... and the per-client bf16 path (autograd over the single-client kernels)
        module = tr.modules[k].train()
        saved = {n: b.clone() for n, b in module.named_buffers()}
        with torch.autocast("cuda", dtype=BF):
            lb = F.cross_entropy(module(x.contiguous(memory_format=torch.channels_last)).float(), y)
        mp = dict(module.named_parameters())
        gbf = dict(zip(mp, torch.autograd.grad(lb, list(mp.values()))))
        for n, b in module.named_buffers():
            b.copy_(saved[n])
        assert float(loss[k]) == pytest.approx(float(l), rel=1e-2)
        assert float(loss[k]) == pytest.approx(float(lb), rel=1e-3)
        for name in params:
            got = tr.layout.view_of(gb[k], tr.layout[name]).float()
            # bf16 rounding noise grows towards the first layers: both bf16 paths sit at
            # 0.93 - 0.95 against fp32 there, and at 0.99 against each other
            assert _cos(got, g32[name]) > 0.9, (k, name, _cos(got, g32[name]))
            assert _cos(got, gbf[name]) > 0.98, (k, name, _cos(got, gbf[name]))
            assert 0.9 < float(got.norm() / g32[name].norm()) < 1.1, name


def test_round_tracks_the_per_client_trainer_and_learns():
    a = _engine("torch", samples=96)
    b = _engine("native", samples=96)
    ra, rb = a.run(1), b.run(1)
    assert ra[0].status == rb[0].status == "COMPLETED"
    assert rb[0].agg_metrics["loss"] == pytest.approx(ra[0].agg_metrics["loss"], rel=0.1)
    ga, gb = a.global_arena.data.float(), b.global_arena.data.float()
    assert _rel(gb, ga) < 0.05
    sd = b.global_state_dict()
    assert int(sd["bn1.num_batches_tracked"]) == 3
    torch.testing.assert_close(sd["bn1.running_mean"], a.global_state_dict()["bn1.running_mean"],
                               rtol=5e-2, atol=5e-3)
    more = b.run(4)
    assert all(r.status == "COMPLETED" for r in more)
    assert more[-1].agg_metrics["loss"] < rb[0].agg_metrics["loss"]
    assert "accuracy" in b.evaluate()


def test_graph_replays_equal_the_eager_steps_bit_for_bit():
    a = _engine("native", graph=False, samples=128)
    b = _engine("native", graph=True, samples=128)
    a.run(2)
    b.run(2)
    assert len(b.trainer._ggraphs) == 1
    assert torch.equal(a.global_arena.data, b.global_arena.data)
    assert int(b.global_state_dict()["bn1.num_batches_tracked"]) == 8  # 4 rounds of 2 steps


def test_unequal_shards_and_ragged_tails():
    """Clients with 96, 40 and 70 samples at batch 32: steps with three, two and one
    full batches, and ragged tails through the per-client path."""
    eng = _engine("native", clients=3, samples=[96, 40, 70])
    reps = eng.run(2)
    assert all(r.status == "COMPLETED" for r in reps)
    assert reps[0].num_clients == 3
    assert torch.isfinite(eng.global_arena.data).all()
    keys = set(eng.trainer._ggraphs)
    assert (0, 3) in keys and len(keys) >= 2


def test_faults_and_rollback_leave_the_trainer_consistent(tmp_path):
    """A dropped client, a poisoned update and a failed round (rolled back to the last
    checkpoint) on the grouped trainer: the run ends with the same model as a run in
    which only the drop and the poison happen."""
    from nanofed_amd.engine import FaultPlan

    def run(fail: bool, base):
        torch.cuda.set_device(0)
        cfg = EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=3,
                           samples_per_client=64, batch_size=32, local_epochs=1, eval_samples=64,
                           learning_rate=0.05, seed=9, backend="native", checkpoint_rounds=True,
                           max_update_norm=1000.0)
        faults = FaultPlan(drop={0: {2}}, poison={1: {0}},
                           fail={2: "HIP error: injected"} if fail else {})
        eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)),
                              base_dir=base, faults=faults)
        reps = eng.run(4)
        eng.flush()
        return eng, reps

    a, ra = run(False, tmp_path / "a")
    b, rb = run(True, tmp_path / "b")
    assert [r.status for r in ra] == ["COMPLETED"] * 4
    assert [r.status for r in rb if r.status == "COMPLETED"] == ["COMPLETED"] * 4
    assert b.recoveries == 1
    assert ra[0].num_clients == 2 and ra[1].num_clients == 2
    assert torch.equal(a.global_arena.data, b.global_arena.data)
    assert int(b.global_state_dict()["bn1.num_batches_tracked"]) == 8  # 4 rounds of 2 steps


def test_groups_larger_than_the_launch_limit_are_stepped_in_chunks(monkeypatch):
    """Three clients with at most two per launch: chunks (0, 2) and (2, 1) must train the
    models of the single launch of three up to bf16 noise (the group size picks the tile)."""
    from nanofed_amd.engine import resnet_native

    whole = _engine("native", clients=3, samples=64)
    ra = whole.run(1)
    monkeypatch.setattr(resnet_native, "MAX_GROUP", 2)
    split = _engine("native", clients=3, samples=64)
    rb = split.run(1)
    assert set(split.trainer._ggraphs) == {(0, 2), (2, 1)}
    assert set(whole.trainer._ggraphs) == {(0, 3)}
    assert rb[0].agg_metrics["loss"] == pytest.approx(ra[0].agg_metrics["loss"], rel=2e-2)
    assert _rel(split.global_arena.data, whole.global_arena.data) < 0.02
