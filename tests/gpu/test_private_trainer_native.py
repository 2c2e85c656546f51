"""``PrivateTrainer.train_batch`` on the native DP-SGD kernels against the ``torch.func``
per-sample step of the same trainer (reference semantics: nanofed/trainer/private.py:54-134,
with per-example clipping instead of its batch clip)."""

import time

import pytest
import torch

from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.models import MNISTModel
from nanofed_amd.privacy.accountant import GaussianAccountant
from nanofed_amd.privacy.config import PrivacyConfig
from nanofed_amd.trainer import PrivateTrainer, TrainingConfig

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(n=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 1, 28, 28, generator=g)
    y = torch.randint(0, 10, (n,), generator=g)
    return x.to(DEV), y.to(DEV)


def _trainer(native, sigma=1e-4, clip=0.5):
    cfg = TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1, device=DEV)
    pc = PrivacyConfig(epsilon=10.0, delta=1e-5, max_gradient_norm=clip, noise_multiplier=sigma)
    return PrivateTrainer(cfg, pc, clip_mode="per_sample", native=native)


def _model(seed=1):
    torch.manual_seed(seed)
    return MNISTModel().to(DEV).eval()  # no dropout: both paths see the same network


@pytest.mark.parametrize("n", [64, 40])
def test_native_step_matches_the_per_sample_oracle(n):
    """Almost no noise: the update is lr times the mean of the clipped per-example
    gradients.  The native step (bf16 compute) must move every tensor as the fp32
    torch.func step does."""
    x, y = _batch(n)
    ref_model, nat_model = _model(), _model()
    before = {k: v.clone() for k, v in ref_model.state_dict().items()}
    ref, nat = _trainer(False), _trainer(True)
    m_ref = ref.train_batch(ref_model, (x, y), torch.optim.SGD(ref_model.parameters(), lr=0.1))
    m_nat = nat.train_batch(nat_model, (x, y), torch.optim.SGD(nat_model.parameters(), lr=0.1))
    assert m_ref.backend == "torch" and m_nat.backend == "native" and nat.last_backend == "native"
    assert m_nat.samples_processed == n
    assert m_nat.loss == pytest.approx(m_ref.loss, rel=2e-2)
    assert m_nat.accuracy == pytest.approx(m_ref.accuracy, abs=2 / n)
    errors = {}
    for k in before:
        d_ref = ref_model.state_dict()[k] - before[k]
        d_nat = nat_model.state_dict()[k] - before[k]
        scale = float(d_ref.abs().max())
        assert scale > 0
        # every example is clipped here (norms are well above C = 0.5), so the deltas test
        # the clip factors as well as the gradients
        rel = float((d_nat - d_ref).norm() / d_ref.norm())
        cos = float(torch.nn.functional.cosine_similarity(d_nat.flatten(), d_ref.flatten(), dim=0))
        errors[k] = (round(rel, 4), round(cos, 5))
        # the bounds of the engine's own DP test (tests/gpu/test_mnist_native_dp.py): clip
        # factors from bf16 activations agree with the fp32 ones to a few percent
        assert cos > 0.99, (k, errors[k])
        assert float(d_nat.norm()) == pytest.approx(float(d_ref.norm()), rel=8e-2), k
    print("parameter change, native bf16 step vs fp32 oracle (relative error, cosine):", errors)
    # the privacy accounting does not depend on the route
    assert nat.get_privacy_spent() == ref.get_privacy_spent()


def test_noise_has_the_configured_scale_and_steps_differ():
    """sigma * C / n on the mean gradient: with lr 1 and a frozen input the parameter change
    of two trainers differs by noise of std sqrt(2) * sigma * C / n."""
    x, y = _batch(64)
    sigma, clip = 2.0, 0.5
    deltas = []
    for _ in range(2):
        model = _model()
        before = model.fc1.weight.detach().clone()
        tr = _trainer(True, sigma=sigma, clip=clip)
        tr.train_batch(model, (x, y), torch.optim.SGD(model.parameters(), lr=1.0))
        deltas.append(model.fc1.weight.detach() - before)
    diff = deltas[0] - deltas[1]
    want = (2.0 ** 0.5) * sigma * clip / 64
    assert float(diff.std()) == pytest.approx(want, rel=0.05)


def test_weights_written_by_somebody_else_are_picked_up():
    x, y = _batch(64)
    model = _model()
    tr = _trainer(True)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    tr.train_batch(model, (x, y), opt)
    with torch.no_grad():
        for p in model.parameters():
            p.zero_()  # e.g. a new global model loaded between two steps
    tr.train_batch(model, (x, y), opt)
    fresh = MNISTModel().to(DEV).eval()
    with torch.no_grad():
        for p in fresh.parameters():
            p.zero_()
    tr2 = _trainer(True)
    tr2.train_batch(fresh, (x, y), torch.optim.SGD(fresh.parameters(), lr=0.1))
    for a, b in zip(model.parameters(), fresh.parameters()):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-5)


def test_calls_the_kernels_do_not_cover_take_the_torch_path_or_raise():
    x, y = _batch(64)
    model = _model()
    adam = torch.optim.Adam(model.parameters(), lr=0.01)
    auto = _trainer("auto")
    assert auto.train_batch(model, (x, y), adam).backend == "torch"
    with pytest.raises(DeviceError, match="plain SGD"):
        _trainer(True).train_batch(model, (x, y), adam)
    cfg = TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1, device=DEV)
    pc = PrivacyConfig(epsilon=10.0, delta=1e-5, max_gradient_norm=1.0, noise_multiplier=1.0)
    batch_clip = PrivateTrainer(cfg, pc, accountant=GaussianAccountant(pc))  # reference mode
    sgd = torch.optim.SGD(model.parameters(), lr=0.1)
    assert batch_clip.train_batch(model, (x, y), sgd).backend == "torch"
    # on a device the default noise generator is the device one (no host randn + copy)
    assert type(batch_clip._noise_gen).__name__ == "DeviceGaussianNoiseGenerator"


def test_native_dp_steps_are_much_faster_than_the_vmap_steps():
    x, y = _batch(64)
    rates = {}
    for native in (False, True):
        model = _model()
        tr = _trainer(native, sigma=1.0)
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        for _ in range(3):
            tr.train_batch(model, (x, y), opt)
        torch.cuda.synchronize()
        steps = 30 if native else 6
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_batch(model, (x, y), opt)
        torch.cuda.synchronize()
        rates[native] = steps / (time.perf_counter() - t0)
    print(f"DP-SGD steps/s, one train_batch call per step: torch.func {rates[False]:.1f}, "
          f"native {rates[True]:.1f}")
    assert rates[True] > 5 * rates[False]


def test_native_dp_epoch_runs_as_one_graph_and_accounts_every_batch():
    from torch.utils.data import DataLoader

    from nanofed_amd.data.mnist import MNISTTensorDataset
    from nanofed_amd.data.synthetic import synthetic_mnist

    images, labels = synthetic_mnist(64 * 40 + 17, seed=3)
    loader = DataLoader(MNISTTensorDataset(images, labels), batch_size=64, shuffle=True)
    steps = len(loader)
    rates, eps = {}, {}
    for native in (False, True):
        torch.manual_seed(0)
        model = MNISTModel().to(DEV)
        tr = _trainer(native, sigma=1.0, clip=1.0)
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        first = tr.train_epoch(model, loader, opt, 0)  # builds the session / warms up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = tr.train_epoch(model, loader, opt, 1)
        torch.cuda.synchronize()
        rates[native] = steps / (time.perf_counter() - t0)
        eps[native] = tr.get_privacy_spent()
        assert m.backend == ("native" if native else "torch") and m.samples_processed == len(images)
        assert m.loss < first.loss  # it trains, clipped and noisy as it is
        assert all(torch.isfinite(p).all() for p in model.parameters())
    # 2 epochs x 41 batches, the last one of 17 samples, accounted for on both routes
    assert eps[True] == eps[False]
    print(f"DP-SGD steps/s inside train_epoch: torch.func {rates[False]:.1f}, native {rates[True]:.1f}")
    assert rates[True] > 20 * rates[False]
