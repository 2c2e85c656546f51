"""Flat arenas and the CPU implementations of the arena ops (the numerical
oracles of the HIP kernels)."""


import pytest
import torch
import torch.nn as nn

from nanofed_amd import ops
from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.engine.mnist_native import make_layout, native_permutations
from nanofed_amd.models import MNISTModel, ResNet18
from nanofed_amd.runtime import ArenaLayout, ArenaPool, ArenaStateDict, ParameterArena


def test_layout_offsets_alignment_and_sizes():
    model = MNISTModel()
    layout = ArenaLayout.from_module(model)
    assert layout.payload_numel == 1_199_882 and layout.numel == 1_200_000
    assert layout.trainable_numel == layout.numel
    assert all(s.offset % 64 == 0 for s in layout.segments)
    assert layout.keys() == list(model.state_dict().keys())
    assert layout["fc1.weight"].shape == (128, 9216)


def test_state_dict_round_trip_and_views_alias_the_buffer():
    model = MNISTModel()
    arena = ParameterArena.from_module(model)
    out = arena.state_dict()
    assert all(torch.equal(out[k], v) for k, v in model.state_dict().items())
    views = arena.views()
    assert isinstance(views, ArenaStateDict) and views.flat is arena.data
    views["fc2.bias"].fill_(3.0)
    seg = arena.layout["fc2.bias"]
    assert torch.equal(arena.data[seg.offset : seg.offset + 10], torch.full((10,), 3.0))
    with pytest.raises(DeviceError):
        arena.load_state_dict({"conv1.weight": torch.zeros(32, 1, 3, 3)})


def test_bind_makes_autograd_and_arena_share_memory():
    model = nn.Sequential(nn.Linear(4, 3), nn.Tanh(), nn.Linear(3, 2))
    arena = ParameterArena.from_module(model, with_grad=True)
    arena.bind(model)
    model(torch.randn(5, 4)).sum().backward()
    assert float(arena.grad.abs().sum()) > 0
    before = arena.data.clone()
    ops.sgd_step(arena.trainable, arena.grad, lr=0.1)
    assert float(arena.grad.abs().sum()) == 0.0
    assert not torch.equal(before, arena.data)
    assert torch.equal(model[0].weight.data, arena.views()["0.weight"])


def test_integer_buffers_and_float_buffers_of_batchnorm_models():
    model = ResNet18(width=8)
    layout = ArenaLayout.from_module(model)
    assert layout.trainable_numel < layout.numel  # buffers follow the parameters
    arena = ParameterArena.from_module(model, with_grad=True)
    arena.bind(model)
    model.train()
    model(torch.randn(2, 3, 32, 32))
    arena.pull_int_buffers(model)
    assert float(arena.views()["bn1.num_batches_tracked"]) == 1.0
    exported = arena.state_dict()
    assert exported["bn1.num_batches_tracked"].dtype == torch.int64
    assert torch.equal(exported["bn1.running_mean"], model.bn1.running_mean)


def test_native_layout_permutations_round_trip():
    model = MNISTModel()
    layout = make_layout(model)
    arena = ParameterArena(layout)
    arena.load_state_dict(model.state_dict())
    v = arena.views()
    assert torch.equal(v["conv2.weight"].reshape(64, 3, 3, 32).permute(0, 3, 1, 2),
                       model.conv2.weight.data)
    assert torch.equal(v["fc1.weight"].reshape(128, 144, 64).permute(0, 2, 1).reshape(128, 9216),
                       model.fc1.weight.data)
    back = arena.state_dict()
    assert all(torch.equal(back[k], t) for k, t in model.state_dict().items())
    with pytest.raises(DeviceError):
        arena.bind(model)
    perms = native_permutations()
    assert sorted(perms["fc1.weight"].tolist()) == list(range(128 * 9216))


def test_pool_broadcast_and_rows():
    layout = ArenaLayout.from_module(nn.Linear(5, 5))
    pool = ArenaPool(layout, 3, "cpu")
    flat = torch.arange(layout.numel, dtype=torch.float32)
    pool.broadcast_from(flat)
    assert all(torch.equal(r, flat) for r in pool.rows())
    pool[1].data.mul_(2)
    assert torch.equal(pool.data[1], 2 * flat) and torch.equal(pool.data[0], flat)


# -- ops (CPU implementations) ------------------------------------------------
def test_weighted_sum_and_zero_weight_exclusion():
    a, b, c = torch.ones(10), torch.full((10,), 2.0), torch.full((10,), float("nan"))
    out = ops.weighted_sum([a, b, c], [0.25, 0.75, 0.0])
    assert torch.equal(out, torch.full((10,), 1.75))
    ops.weighted_sum([a], torch.tensor([2.0]), out=out, accumulate=True)
    assert torch.equal(out, torch.full((10,), 3.75))
    with pytest.raises(DeviceError):
        ops.weighted_sum([a, torch.ones(3)], [0.5, 0.5])


def test_sqnorm_finite_and_clip_noise():
    x = torch.tensor([3.0, 4.0, float("nan"), float("inf")])
    sq, bad = ops.finite_and_sqnorm(x)
    assert (sq, bad) == (25.0, 2)
    y = torch.full((100,), 1.0)
    out = ops.clip_noise(y, max_norm=1.0, noise_std=0.0, seed=0, offset=0)
    assert float(out.norm()) == pytest.approx(1.0, rel=1e-5)
    n1 = ops.clip_noise(y, 1.0, 0.3, seed=5, offset=2)
    n2 = ops.clip_noise(y, 1.0, 0.3, seed=5, offset=2)
    assert torch.equal(n1, n2)
    big = ops.clip_noise(torch.zeros(200_000), 1.0, 0.3, seed=5, offset=3)
    assert float(big.std()) == pytest.approx(0.3, rel=0.02)


def test_philox_host_stream_properties():
    n = 100_000
    u = ops.philox_fill(torch.empty(n), seed=1, offset=0, mode=ops.MODE_UNIFORM)
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0
    assert float(u.mean()) == pytest.approx(0.5, abs=0.01)
    z = ops.philox_fill(torch.empty(n), seed=1, offset=0, mode=ops.MODE_NORMAL)
    assert float(z.std()) == pytest.approx(1.0, abs=0.02)
    lap = ops.philox_fill(torch.empty(n), seed=1, offset=0, mode=ops.MODE_LAPLACE)
    assert float(lap.abs().mean()) == pytest.approx(1.0, abs=0.03)
    # element i depends only on (seed, offset, i)
    whole = ops.philox_fill(torch.empty(1001), seed=3, offset=9)
    parts = torch.empty(1001)
    ops.philox_fill(parts[:333], seed=3, offset=9, first=0)
    ops.philox_fill(parts[333:], seed=3, offset=9, first=333)
    assert torch.equal(whole, parts)
    other = ops.philox_fill(torch.empty(1001), seed=3, offset=10)
    assert not torch.equal(whole, other)


def test_sgd_step_and_scale_accumulate():
    p, g = torch.ones(8), torch.full((8,), 0.5)
    shadow = torch.zeros(8, dtype=torch.bfloat16)
    ops.sgd_step(p, g, lr=0.2, weight_decay=0.1, zero_grad=True, bf16_shadow=shadow)
    assert torch.allclose(p, torch.full((8,), 1 - 0.2 * (0.5 + 0.1)))
    assert float(g.abs().sum()) == 0 and torch.equal(shadow, p.to(torch.bfloat16))
    out = torch.ones(4)
    ops.scale_accumulate(out, torch.full((4,), 2.0), 0.5, 3.0)
    assert torch.equal(out, torch.full((4,), 6.5))
    a, b = torch.zeros(64), torch.zeros(64)
    ops.mask_add_(a, seed=11, scale=2.0)
    ops.mask_add_(b, seed=11, scale=-2.0)
    assert float(a.abs().max()) > 0 and float((a + b).abs().max()) == 0.0


def test_channels_last_storage_keeps_values_binding_and_export():
    import torch.nn as nn

    from nanofed_amd.runtime.arena import ArenaLayout, ParameterArena

    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(),
                        nn.Conv2d(8, 4, 1), nn.Flatten(), nn.Linear(4 * 6 * 6, 5))
    net.eval()  # keep the running statistics fixed
    want = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(2, 3, 6, 6)
    y_ref = net(x).detach()

    layout = ArenaLayout.from_module(net).with_channels_last()
    assert set(layout.perms) == {"0.weight", "3.weight"} and layout.bindable()
    arena = ParameterArena(layout, "cpu", with_grad=True)
    arena.load_state_dict(want)
    views = arena.views()
    for k, v in want.items():
        assert torch.equal(views[k].to(v.dtype), v), k
    w = views["0.weight"]
    assert w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous()
    # stored order is [O, H, W, I]
    seg = layout["0.weight"]
    stored = arena.data[seg.offset : seg.offset + seg.numel].view(8, 3, 3, 3)
    assert torch.equal(stored, want["0.weight"].permute(0, 2, 3, 1))

    bound = arena.bind(net)
    torch.testing.assert_close(bound(x), y_ref)
    bound(x).sum().backward()
    g = arena.grad_views()["0.weight"]
    assert g.data_ptr() == dict(bound.named_parameters())["0.weight"].grad.data_ptr()
    assert float(g.abs().sum()) > 0
    exported = arena.state_dict()
    for k, v in want.items():
        assert torch.equal(exported[k], v), k
    # another arena with the plain layout receives the same model through the state dict
    plain = ParameterArena(ArenaLayout.from_module(net), "cpu")
    plain.load_state_dict(arena.views())
    assert torch.equal(plain.state_dict()["3.weight"], want["3.weight"])


def test_sgd_multi_matches_per_tensor_sgd_on_cpu():
    from nanofed_amd import ops

    torch.manual_seed(3)
    params = torch.randn(64 + 128 + 16)
    ref = params.clone()
    shadow = torch.zeros(params.numel(), dtype=torch.bfloat16)
    g0 = torch.randn(4, 4, 2, 2).contiguous(memory_format=torch.channels_last)  # 64, strided
    g1 = torch.randn(128).to(torch.bfloat16)
    ops.sgd_multi(params, [g0, g1], [0, 64], lr=0.1, weight_decay=0.01, bf16_shadow=shadow)
    # storage order of the channels-last gradient is [O, H, W, I]
    flat0 = g0.permute(0, 2, 3, 1).reshape(-1)
    exp0 = ref[:64] - 0.1 * (flat0 + 0.01 * ref[:64])
    exp1 = ref[64:192] - 0.1 * (g1.float() + 0.01 * ref[64:192])
    assert torch.allclose(params[:64], exp0)
    assert torch.allclose(params[64:192], exp1)
    assert torch.equal(params[192:], ref[192:])
    assert torch.equal(shadow[:192], params[:192].to(torch.bfloat16))
    with pytest.raises(Exception):
        ops.sgd_multi(params, [g1], [150], lr=0.1)
