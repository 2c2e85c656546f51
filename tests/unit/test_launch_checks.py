"""Host-side operand checks of the raw-address launchers (``ops.group_ops``,
``ops.gemm_ops._launch``): a shape slip in a caller must be a ``DeviceError`` on the host,
never an out-of-bounds access on the device.  Every case here is rejected before a launch,
so host tensors stand in for device tensors."""

import pytest
import torch

from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.ops import gemm_ops
from nanofed_amd.ops import group_ops as go
from nanofed_amd.ops.group_ops import Ref

BF, F32 = torch.bfloat16, torch.float32


@pytest.fixture(autouse=True)
def _admit_host_tensors(monkeypatch):
    monkeypatch.setattr(go, "_DEVICE_TYPES", ("cuda", "cpu"))
    monkeypatch.setattr(gemm_ops, "_DEVICE_TYPES", ("cuda", "cpu"))


def z(*shape, dtype=BF):
    return torch.zeros(*shape, dtype=dtype)


def test_device_tensors_are_required_outside_these_tests(monkeypatch):
    monkeypatch.setattr(go, "_DEVICE_TYPES", ("cuda",))
    with pytest.raises(DeviceError, match="device tensor"):
        go.im2col(z(4 * 8 * 8 * 8), z(4 * 4 * 4 * 72), 4, 8, 8, 8, 4, 4, 3, 2, 1, 72)


def test_conv3x3_operands():
    G, B, H, W, ci, co = 2, 64, 32, 32, 64, 64
    x, y = z(G * B * H * W * ci), z(G * B * H * W * co)
    w = Ref(z(G, 9 * ci * co + 128), 128)
    with pytest.raises(DeviceError, match="not covered"):
        go.conv3x3_fwd(x, w, y, None, G, B, H, W, 48, co)       # channel count the kernels lack
    with pytest.raises(DeviceError, match="touches"):
        go.conv3x3_fwd(x[:-8].clone(), w, y, None, G, B, H, W, ci, co)  # input one piece short
    with pytest.raises(DeviceError, match="outside a row"):
        go.conv3x3_fwd(x, Ref(w.base, 256), y, None, G, B, H, W, ci, co)
    with pytest.raises(DeviceError, match="rows for"):
        go.conv3x3_fwd(x, Ref(w.base[:1], 128), y, None, G, B, H, W, ci, co)
    with pytest.raises(DeviceError, match="dtype"):
        go.conv3x3_dgrad(y.float(), w, x, G, B, H, W, ci, co)
    with pytest.raises(DeviceError, match="workspace"):
        go.conv3x3_wgrad(x, y, w, z(16), G, B, H, W, ci, co)
    with pytest.raises(DeviceError, match="colsum"):
        go.conv3x3_fwd(x, w, y, z(8, dtype=F32), G, B, H, W, ci, co)


def test_batchnorm_operands():
    G, rows, C = 2, 4096, 64
    x = z(G * rows * C)
    arena = z(G, 4 * C, dtype=F32)
    gamma, beta = Ref(arena, 0), Ref(arena, C)
    stats, ws = z(G * 2 * C, dtype=F32), z(go.bn_workspace_floats(G, C), dtype=F32)
    with pytest.raises(DeviceError, match="stats"):
        go.bn_fwd(x, None, x.clone(), gamma, beta, None, None, stats[:-1], ws, G, rows, C, True, 0.1, 1e-5)
    with pytest.raises(DeviceError, match="workspace"):
        go.bn_fwd(x, None, x.clone(), gamma, beta, None, None, stats, ws[:100], G, rows, C, True, 0.1, 1e-5)
    with pytest.raises(DeviceError, match="multiple of 8"):
        go.bn_fwd(x, None, x.clone(), gamma, beta, None, None, stats, ws, G, rows, 60, True, 0.1, 1e-5)
    with pytest.raises(DeviceError, match="beta"):
        go.bn_fwd(x, None, x.clone(), gamma, Ref(arena, 4 * C - 8), None, None, stats, ws, G, rows, C,
                  True, 0.1, 1e-5)
    gb = z(G, 4 * C)
    with pytest.raises(DeviceError, match="dx"):
        go.bn_bwd(x, None, x, x, stats, gamma, x[:64], None, Ref(gb, 0), Ref(gb, C), ws, G, rows, C, True)
    with pytest.raises(DeviceError, match="same buffer"):
        go.bn_bwd(x, None, x, x, stats, gamma, x.clone(), None, Ref(gb, 0), Ref(gb.float(), C), ws,
                  G, rows, C, True)


def test_fused_batchnorm_backward_operands():
    """The data gradient / scatter with the BatchNorm backward epilogue, and the BatchNorm
    pass that takes their partial sums."""
    G, B, H, W, ci, co = 2, 64, 32, 32, 64, 64
    n = G * B * H * W
    dy, dx, act = z(n * co), z(n * ci), z(n * ci)
    w = Ref(z(G, 9 * ci * co), 0)
    arena = z(G, 4 * ci, dtype=F32)
    stats = z(G * 2 * ci, dtype=F32)
    # (what a real caller sizes with conv3x3_tiles; here any count that is clearly too small)
    small = z(8, dtype=F32)
    with pytest.raises(DeviceError, match="colsum"):
        go.conv3x3_dgrad(dy, w, dx, G, B, H, W, ci, co, fuse=go.BnBwdFuse(act, act, None, small))
    big = z(G * 1024 * 2 * ci, dtype=F32)
    with pytest.raises(DeviceError, match="xpre"):
        go.conv3x3_dgrad(dy, w, dx, G, B, H, W, ci, co, fuse=go.BnBwdFuse(act, act[:-8].clone(), None, big))
    with pytest.raises(DeviceError, match="alias"):
        go.conv3x3_dgrad(dy, w, dx, G, B, H, W, ci, co, fuse=go.BnBwdFuse(act, act, dx, big))
    with pytest.raises(DeviceError, match="stats"):   # no mask: the statistics are needed
        go.conv3x3_dgrad(dy, w, dx, G, B, H, W, ci, co, fuse=go.BnBwdFuse(None, act, None, big))
    with pytest.raises(DeviceError, match="beta"):
        go.conv3x3_dgrad(dy, w, dx, G, B, H, W, ci, co,
                         fuse=go.BnBwdFuse(None, act, None, big, stats, Ref(arena, 0), Ref(arena, 4 * ci - 8)))
    rows = B * H * W
    ws = z(go.bn_workspace_floats(G, ci), dtype=F32)
    gb = z(G, 4 * ci)
    with pytest.raises(DeviceError, match="partial sums"):
        go.bn_bwd_pre(act, act, stats, Ref(arena, 0), dx, Ref(gb, 0), Ref(gb, ci), small, 256, ws, G,
                      rows, ci)
    with pytest.raises(DeviceError, match="pre_blocks"):
        go.bn_bwd_pre(act, act, stats, Ref(arena, 0), dx, Ref(gb, 0), Ref(gb, ci), big, 0, ws, G,
                      rows, ci)
    # scatter of a 1x1 / stride 2 convolution, 16x16 -> 8x8 per image
    Bs, Hs, C = 8, 16, 64
    col = z(G * Bs * 8 * 8 * C)
    t = z(G * Bs * Hs * Hs * C)
    with pytest.raises(DeviceError, match="mask"):
        go.col2im_join(col, t.clone(), None, t[:-8].clone(), t, big, G, Bs, Hs, Hs, C, 8, 8, 1, 2, 0, C)
    with pytest.raises(DeviceError, match="must divide 256"):
        go.col2im_join(z(G * Bs * 8 * 8 * 48), z(G * Bs * Hs * Hs * 48), None, z(G * Bs * Hs * Hs * 48),
                       z(G * Bs * Hs * Hs * 48), big, G, Bs, Hs, Hs, 48, 8, 8, 1, 2, 0, 48)
    with pytest.raises(DeviceError, match="alias"):
        go.col2im_join(col, t, t, t.clone(), t.clone(), big, G, Bs, Hs, Hs, C, 8, 8, 1, 2, 0, C)


def test_batched_gemm_operands():
    G, M, N, K = 2, 128, 64, 72
    a, b, c = z(G, M * K), z(G, N * K), z(G, M * N)
    stream = None
    ok = dict(slabs=z(G * 64 * M * N, dtype=F32), stream=stream)
    with pytest.raises(DeviceError, match="raw addresses"):
        go.gemm_batched(gemm_ops.NT, a.data_ptr(), b, c, M, N, K, K, K, N, G, M * K, N * K, M * N, **ok)
    with pytest.raises(DeviceError, match="A: .*touches"):
        go.gemm_batched(gemm_ops.NT, z(G * M * K - 8), b, c, M, N, K, K, K, N, G, M * K, N * K, M * N, **ok)
    with pytest.raises(DeviceError, match="C: .*touches"):
        go.gemm_batched(gemm_ops.NT, a, b, c, M, N, K, K, K, N, G, M * K, N * K, 2 * M * N, **ok)
    with pytest.raises(DeviceError, match="row pitches"):
        go.gemm_batched(gemm_ops.NT, a, b, c, M, N, K, K - 8, K, N, G, M * K, N * K, M * N, **ok)
    with pytest.raises(DeviceError, match="row pitches"):
        go.gemm_batched(gemm_ops.NT, a, b, z(G, M * 60), M, 60, K, K, K, 60, G, M * K, N * K, M * 60, **ok)
    # the weight as a Ref into an arena row that is too short for N x K
    with pytest.raises(DeviceError, match="B: .*touches"):
        go.gemm_batched(gemm_ops.NT, a, Ref(z(G, N * K), 64), c, M, N, K, K, K, N, G, M * K, N * K,
                        M * N, **ok)
    with pytest.raises(DeviceError, match="bias"):
        go.gemm_batched(gemm_ops.NT, a, b, c, M, N, K, K, K, N, G, M * K, N * K, M * N,
                        bias=Ref(z(G, 32), 0), **ok)
    with pytest.raises(DeviceError, match="split-K"):
        go.gemm_batched(gemm_ops.TN, z(G, 4096 * 64), z(G, 4096 * 64), z(G, 64 * 64), 64, 64, 4096,
                        64, 64, 64, G, 4096 * 64, 4096 * 64, 64 * 64, z(8, dtype=F32), stream)


def test_patch_gather_and_head_operands():
    with pytest.raises(DeviceError, match="output size"):
        go.im2col(z(4 * 8 * 8 * 8), z(4 * 4 * 4 * 72), 4, 8, 8, 8, 5, 4, 3, 2, 1, 72)
    with pytest.raises(DeviceError, match="patch matrix"):
        go.im2col(z(4 * 8 * 8 * 8), z(4 * 4 * 4 * 72 - 8), 4, 8, 8, 8, 4, 4, 3, 2, 1, 72)
    with pytest.raises(DeviceError, match="padded patch length"):
        go.col2im(z(4 * 4 * 4 * 64), z(4 * 8 * 8 * 8), 4, 8, 8, 8, 4, 4, 3, 2, 1, 64)
    G, B, px, C, classes = 2, 64, 16, 512, 10
    arena = z(G, classes * C + classes + 64)
    with pytest.raises(DeviceError, match="labels"):
        go.pool_fc_ce(z(G * B * px * C), Ref(arena, 0), Ref(arena, classes * C), z(G * B - 1, dtype=torch.int64),
                      z(G * B * C), z(G * B * classes), z(G * B, dtype=F32), z(G * B, dtype=F32), None,
                      G, B, px, C, classes)
    with pytest.raises(DeviceError, match="sgd_rows"):
        go.sgd_rows(z(2, 64, dtype=F32), z(2, 32), z(2, 64), 64, 0.1, 0.0)


def test_single_gemm_operand_shapes_and_padding():
    M, N, K = 64, 40, 128
    x, w = z(M, K), z(N, K)
    out = z(M, N)
    with pytest.raises(DeviceError, match="needs"):
        gemm_ops._launch(gemm_ops.NT, x, z(N, K + 8), out, M, N, K)       # weight with another K
    with pytest.raises(DeviceError, match="needs"):
        gemm_ops._launch(gemm_ops.NN, x, w, z(M, K), M, K, N)            # (N, K) where (K, N) is due
    with pytest.raises(DeviceError, match="expected"):
        gemm_ops._launch(gemm_ops.NT, x, w, z(M, N + 8), M, N, K)
    # N = 36: the last 16-byte piece of a row spills into columns 36..39.  Fine for rows that
    # start at column 0 of a padded buffer, not for a slice in the middle of a wider matrix
    wide = z(M, 80)
    with pytest.raises(DeviceError, match="column slice"):
        gemm_ops._launch(gemm_ops.NT, x, z(36, K), wide[:, 40:76], M, 36, K)
