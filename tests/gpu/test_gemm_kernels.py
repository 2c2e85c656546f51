"""The hand-written bf16 MFMA GEMM (csrc/hip/gemm_ops.hip) in its three operand
layouts against fp32 ``torch.mm`` of the same bf16 inputs: the GPT-2-small layer
shapes at 4096 rows, ragged shapes (M, N, K not multiples of the tile, K tail inside
a 16-byte piece, padded leading dimensions holding NaNs), the bias / GELU epilogue
and the fixed-order split-K weight gradient."""

import pytest
import torch

from nanofed_amd.ops import gemm_ops, nn_ops

pytestmark = pytest.mark.gpu


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


def close(got, want, k):
    # bf16 result of a length-k fp32 accumulation: half an ulp of the result plus
    # accumulated product rounding
    err = (got.float() - want).abs()
    tol = 8e-3 * want.abs() + 2e-3 * want.abs().max().clamp(min=1e-3)
    assert bool((err <= tol).all()), (float(err.max()), float(want.abs().max()), k)


GPT2_SHAPES = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]  # (N, K) of c_attn, c_proj, c_fc, mlp c_proj


@pytest.mark.parametrize("N,K", GPT2_SHAPES)
def test_gpt2_small_layer_shapes(N, K):
    assert gemm_ops.enabled(torch.device("cuda"))
    M = 4096
    x, w, b = rnd(M, K, seed=1), rnd(N, K, scale=0.05, seed=2), rnd(N, seed=3)
    dy = rnd(M, N, scale=0.1, seed=4)
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_fwd(x, w, b, y)
    close(y, x.float() @ w.float().t() + b.float(), K)
    dx = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_dgrad(dy, w, dx)
    close(dx, dy.float() @ w.float(), N)
    dw = torch.empty(N, K, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_wgrad(dy, x, dw)
    close(dw, dy.float().t() @ x.float(), M)
    again = torch.empty_like(dw)
    gemm_ops.linear_wgrad(dy, x, again)
    assert torch.equal(dw, again)  # split-K slabs are summed in a fixed order


def test_lm_head_shapes_with_padded_logits_holding_nans():
    # d(logits) lives in a buffer whose rows are padded to a multiple of 64; the
    # padding is never written by anybody and must not leak into any result
    R, V, C = 256, 1021, 192
    ld = (V + 63) // 64 * 64
    lnf, wte = rnd(R, C, seed=5), rnd(V, C, scale=0.05, seed=6)
    buf = torch.full((R, ld), float("nan"), dtype=torch.bfloat16, device="cuda")
    logits = buf[:, :V]
    gemm_ops.linear_fwd(lnf, wte, None, logits)
    close(logits, lnf.float() @ wte.float().t(), C)
    dlog = rnd(R, V, scale=0.01, seed=7)
    buf.fill_(float("nan"))
    logits.copy_(dlog)
    dln = torch.empty(R, C, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_dgrad(logits, wte, dln)          # reduction over V = 1021: K tail of 5 elements
    close(dln, dlog.float() @ wte.float(), V)
    dwte = torch.empty(V, C, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_wgrad(logits, lnf, dwte)         # M = V rows of output, not a multiple of 8
    close(dwte, dlog.float().t() @ lnf.float(), R)


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (100, 72, 40), (130, 264, 200), (1, 8, 8), (257, 136, 1000)])
def test_ragged_shapes(M, N, K):
    x, w = rnd(M, K, seed=8), rnd(N, K, seed=9)
    dy = rnd(M, N, seed=10)
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_fwd(x, w, None, y)
    close(y, x.float() @ w.float().t(), K)
    dx = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_dgrad(dy, w, dx)
    close(dx, dy.float() @ w.float(), N)
    dw = torch.empty(N, K, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_wgrad(dy, x, dw)
    close(dw, dy.float().t() @ x.float(), M)


def test_identity_against_an_asymmetric_matrix_catches_a_transposed_store():
    n = 128
    eye = torch.eye(n, device="cuda", dtype=torch.bfloat16)
    asym = (torch.arange(n * n, device="cuda").view(n, n) % 251).to(torch.bfloat16)
    out = torch.empty(n, n, dtype=torch.bfloat16, device="cuda")
    gemm_ops.linear_fwd(eye, asym, None, out)       # I @ asym^T
    assert torch.equal(out, asym.t())
    gemm_ops.linear_dgrad(eye, asym, out)           # I @ asym
    assert torch.equal(out, asym)
    gemm_ops.linear_wgrad(eye, asym, out)           # I^T @ asym
    assert torch.equal(out, asym)


def test_gelu_epilogue_equals_the_separate_kernel():
    M, N, K = 512, 384, 256
    x, w, b = rnd(M, K, seed=11), rnd(N, K, scale=0.1, seed=12), rnd(N, seed=13)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    act = torch.empty_like(pre)
    gemm_ops.linear_fwd(x, w, b, pre, gelu_out=act)
    plain = torch.empty_like(pre)
    gemm_ops.linear_fwd(x, w, b, plain)
    assert torch.equal(pre, plain)
    want = torch.empty_like(pre)
    nn_ops.gelu_fwd(pre, want)
    assert torch.equal(act, want)


def test_operand_checks_fail_loudly():
    from nanofed_amd.core.exceptions import DeviceError

    x = rnd(64, 72)
    with pytest.raises(DeviceError):
        gemm_ops.linear_fwd(x[:, :70], rnd(8, 70), None, torch.empty(64, 8, dtype=torch.bfloat16, device="cuda"))
    with pytest.raises(DeviceError):
        gemm_ops.linear_fwd(x.float(), rnd(8, 72), None, torch.empty(64, 8, dtype=torch.bfloat16, device="cuda"))


@pytest.mark.parametrize("geo", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(300, 520, 200), (1024, 768, 1536), (129, 193, 77)])
def test_every_tile_geometry_on_every_layout(geo, M, N, K):
    from nanofed_amd.core.exceptions import DeviceError

    kp = (K + 7) // 8 * 8

    def padded(t):  # rows padded to a multiple of 8 elements, the padding holding NaNs
        buf = torch.full((t.shape[0], (t.shape[1] + 7) // 8 * 8), float("nan"),
                         dtype=torch.bfloat16, device="cuda")
        buf[:, : t.shape[1]] = t
        return buf[:, : t.shape[1]]

    a, b = padded(rnd(M, K, seed=geo + 1)), padded(rnd(N, K, scale=0.05, seed=geo + 2))
    ld = (N + 7) // 8 * 8
    out = torch.full((M, ld), float("nan"), dtype=torch.bfloat16, device="cuda")[:, :N]
    gemm_ops._launch(gemm_ops.NT, a, b, out, M, N, K, geo=geo, splits=1)
    close(out, a.float() @ b.float().t(), K)
    # the other two layouts: geometry 4 serves k-contiguous operands only
    dy = padded(rnd(M, N, scale=0.1, seed=9))
    dx = torch.empty(M, kp, dtype=torch.bfloat16, device="cuda")[:, :K]
    if geo == 4:
        with pytest.raises(DeviceError):
            gemm_ops._launch(gemm_ops.NN, dy, b, dx, M, K, N, geo=geo, splits=1)
        return
    gemm_ops._launch(gemm_ops.NN, dy, b, dx, M, K, N, geo=geo, splits=1)
    close(dx, dy.float() @ b.float(), N)
    dw = torch.empty(N, kp, dtype=torch.bfloat16, device="cuda")[:, :K]
    gemm_ops._launch(gemm_ops.TN, dy, a, dw, N, K, M, geo=geo, splits=1)
    close(dw, dy.float().t() @ a.float(), M)
