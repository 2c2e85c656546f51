"""HIP arena kernels against fp32 PyTorch references of the same arithmetic."""


import pytest
import torch

from nanofed_amd import ops

pytestmark = pytest.mark.gpu

SIZES = [1, 3, 4, 257, 4096 + 5, 1_199_882]


def _dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("k", [1, 3, 8])
def test_weighted_sum_matches_reference(n, k):
    g = torch.Generator().manual_seed(n * 31 + k)
    srcs = [torch.randn(n, generator=g) for _ in range(k)]
    w = torch.rand(k, generator=g)
    want = sum(wi * s for wi, s in zip(w, srcs))
    got = ops.weighted_sum([s.to(_dev()) for s in srcs], w.to(_dev()))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)


def test_weighted_sum_skips_zero_weight_nan_source():
    a = torch.ones(1000, device=_dev())
    b = torch.full((1000,), float("nan"), device=_dev())
    got = ops.weighted_sum([a, b], [1.0, 0.0])
    assert torch.equal(got.cpu(), torch.ones(1000))


@pytest.mark.parametrize("n", SIZES)
def test_sqnorm_finite(n):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    if n > 3:
        x[1] = float("nan")
        x[n - 1] = float("inf")
    res = ops.sqnorm_finite(x.to(_dev())).cpu()
    ok = torch.isfinite(x)
    want = x[ok].double().square().sum().item()
    assert res[0].item() == pytest.approx(want, rel=1e-4)
    assert res[1].item() == float((~ok).sum())


@pytest.mark.parametrize("n", SIZES)
def test_sgd_step_updates_clears_and_shadows(n):
    g = torch.Generator().manual_seed(n)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    pd, gd = p.to(_dev()), gr.to(_dev())
    shadow = torch.zeros(n, dtype=torch.bfloat16, device=_dev())
    ops.sgd_step(pd, gd, lr=0.1, weight_decay=0.01, zero_grad=True, bf16_shadow=shadow)
    want = p - 0.1 * (gr + 0.01 * p)
    torch.testing.assert_close(pd.cpu(), want, rtol=1e-6, atol=1e-6)
    assert float(gd.abs().max()) == 0.0
    assert torch.equal(shadow.cpu(), pd.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("mode", [ops.MODE_UNIFORM, ops.MODE_NORMAL, ops.MODE_LAPLACE, ops.MODE_SYM_UNIFORM])
def test_philox_device_matches_host_stream(mode):
    n = 10_007
    dev = ops.philox_fill(torch.empty(n, device=_dev()), seed=1234, offset=7, mode=mode)
    host = ops.philox_fill(torch.empty(n), seed=1234, offset=7, mode=mode)
    torch.testing.assert_close(dev.cpu(), host, rtol=2e-4, atol=2e-5)


def test_philox_is_independent_of_call_splitting():
    n = 5001
    whole = ops.philox_fill(torch.empty(n, device=_dev()), seed=5, offset=1)
    parts = torch.empty(n, device=_dev())
    cut = 1237  # not a multiple of 4: exercises the unaligned path
    ops.philox_fill(parts[:cut], seed=5, offset=1, first=0)
    ops.philox_fill(parts[cut:], seed=5, offset=1, first=cut)
    assert torch.equal(whole, parts)


def test_philox_normal_moments():
    x = ops.philox_fill(torch.empty(1 << 20, device=_dev()), seed=9, offset=0)
    assert abs(float(x.mean())) < 5e-3
    assert float(x.std()) == pytest.approx(1.0, abs=5e-3)


def test_clip_noise_clips_and_is_deterministic():
    n = 100_003
    x = torch.randn(n) * 3
    xd = x.to(_dev())
    out = ops.clip_noise(xd, max_norm=1.0, noise_std=0.0, seed=1, offset=0)
    assert float(out.norm()) == pytest.approx(1.0, rel=1e-4)
    torch.testing.assert_close(
        out.cpu(), x * min(1.0, 1.0 / (float(x.norm()) + 1e-6)), rtol=1e-5, atol=1e-6
    )
    a = ops.clip_noise(xd, 1.0, 0.5, seed=3, offset=11)
    b = ops.clip_noise(xd, 1.0, 0.5, seed=3, offset=11)
    c = ops.clip_noise(xd, 1.0, 0.5, seed=3, offset=12)
    assert torch.equal(a, b) and not torch.equal(a, c)
    noise = (a - out).cpu()
    assert float(noise.std()) == pytest.approx(0.5, rel=2e-2)
    host = ops.clip_noise(x, 1.0, 0.5, seed=3, offset=11)
    torch.testing.assert_close(a.cpu(), host, rtol=1e-3, atol=1e-4)


def test_scale_accumulate_and_mask_cancellation():
    n = 70_001
    x = torch.randn(n, device=_dev())
    y = torch.randn(n, device=_dev())
    want = 0.5 * x + 2.0 * y
    ops.scale_accumulate(x, y, 0.5, 2.0)
    torch.testing.assert_close(x, want, rtol=1e-6, atol=1e-6)
    a, b = torch.zeros(n, device=_dev()), torch.zeros(n, device=_dev())
    ops.mask_add_(a, seed=77, scale=+4.0)
    ops.mask_add_(b, seed=77, scale=-4.0)
    assert float(a.abs().max()) > 1.0
    assert float((a + b).abs().max()) == 0.0


def test_arena_kernels_replay_in_hip_graph():
    """A captured sequence of arena kernels gives the eager result on every
    replay (no memset nodes, scalars read from device memory)."""
    n = 300_000
    dev = _dev()
    srcs = [torch.randn(n, device=dev) for _ in range(4)]
    w = torch.tensor([0.1, 0.2, 0.3, 0.4], device=dev)
    out = torch.zeros(n, device=dev)
    res = torch.zeros(2, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.weighted_sum(srcs, w, out=out)
        ops.sqnorm_finite(out, out=res)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.weighted_sum(srcs, w, out=out)
        ops.sqnorm_finite(out, out=res)
    for it in range(4):
        for t in srcs:
            t.normal_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = sum(wi * t for wi, t in zip(w, srcs))
        torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
        assert float(res[0]) == pytest.approx(float(want.double().square().sum()), rel=1e-4)


def test_sgd_multi_matches_host_path():
    """Mixed fp32 / bf16 gradients, one of them channels-last, against the CPU
    implementation of the same op."""
    from nanofed_amd import ops

    gen = torch.Generator().manual_seed(11)
    sizes = [(64, 32, 3, 3), (70,), (128, 64), (3,), (256, 128, 3, 3)]
    offsets, off = [], 0
    for shp in sizes:
        offsets.append(off)
        n = 1
        for d in shp:
            n *= d
        off += (n + 63) // 64 * 64
    params = torch.randn(off, generator=gen)
    grads = []
    for i, shp in enumerate(sizes):
        g = torch.randn(*shp, generator=gen)
        if len(shp) == 4:
            g = g.contiguous(memory_format=torch.channels_last)
        if i % 2 == 0:
            g = g.to(torch.bfloat16)
        grads.append(g)
    host_p, host_s = params.clone(), torch.zeros(off, dtype=torch.bfloat16)
    ops.sgd_multi(host_p, grads, offsets, lr=0.05, weight_decay=0.01, bf16_shadow=host_s)
    dev_p = params.cuda()
    dev_s = torch.zeros(off, dtype=torch.bfloat16, device="cuda")
    ops.sgd_multi(dev_p, [g.cuda() for g in grads], offsets, lr=0.05, weight_decay=0.01,
                  bf16_shadow=dev_s)
    torch.cuda.synchronize()
    assert torch.allclose(dev_p.cpu(), host_p, rtol=1e-6, atol=1e-7)
    touched = torch.zeros(off, dtype=torch.bool)
    for o, g in zip(offsets, grads):
        touched[o : o + g.numel()] = True
    assert torch.equal(dev_s.cpu()[touched], dev_p.cpu().to(torch.bfloat16)[touched])
    assert torch.equal(dev_p.cpu()[~touched], params[~touched])
