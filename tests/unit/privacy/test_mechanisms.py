"""Clip + noise mechanisms (nanofed/privacy/mechanisms.py:17-174)."""

from unittest.mock import Mock

import pytest
import torch

from nanofed_amd.privacy import GaussianNoiseGenerator, PrivacyConfig
from nanofed_amd.privacy.accountant import GaussianAccountant, PrivacySpent
from nanofed_amd.privacy.mechanisms import (
    CentralPrivacyMechanism,
    LocalPrivacyMechanism,
    PrivacyMechanismFactory,
    PrivacyType,
    UpdateMetadata,
)


class RecordingAccountant:
    def __init__(self):
        self.events = []

    def add_noise_event(self, sigma, samples):
        self.events.append((sigma, samples))

    def get_privacy_spent(self):
        return PrivacySpent(0.1 * len(self.events), 1e-5)

    def validate_budget(self, config=None):
        return len(self.events) < 3


def ones_generator():
    gen = Mock(spec=GaussianNoiseGenerator)
    gen.generate.side_effect = lambda shape, scale: torch.ones(shape) * scale
    return gen


CFG = PrivacyConfig(epsilon=1.0, delta=1e-5, max_gradient_norm=1.0, noise_multiplier=2.0)


def small_state():
    return {"w": torch.full((4, 4), 0.01), "b": torch.full((4,), 0.01)}


def test_central_noise_scale_is_sigma_c_over_batch():
    acc, gen = RecordingAccountant(), ones_generator()
    mech = CentralPrivacyMechanism(CFG, accountant=acc, noise_generator=gen)
    state = small_state()
    out = mech.add_noise(state, batch_size=8)
    scale = 2.0 * 1.0 / 8
    for key in state:
        torch.testing.assert_close(out[key], state[key] + scale)
    assert gen.generate.call_count == 2
    for call in gen.generate.call_args_list:
        assert call.kwargs["scale"] == pytest.approx(scale)
    assert acc.events == [(2.0, 8)]
    assert mech.privacy_type is PrivacyType.CENTRAL


def test_local_mechanism_always_uses_batch_one():
    acc, gen = RecordingAccountant(), ones_generator()
    mech = LocalPrivacyMechanism(CFG, accountant=acc, noise_generator=gen)
    out = mech.add_noise(small_state(), batch_size=32)
    torch.testing.assert_close(out["b"], torch.full((4,), 0.01 + 2.0))
    assert acc.events == [(2.0, 1)]
    assert mech.privacy_type is PrivacyType.LOCAL


def test_clipping_uses_the_global_norm_of_the_whole_update():
    mech = CentralPrivacyMechanism(CFG)
    state = {"a": torch.full((9,), 2.0), "b": torch.full((16,), 2.0)}  # norm = 10
    clipped, meta = mech._clip_update(state, max_norm=1.0)
    assert isinstance(meta, UpdateMetadata)
    assert meta.total_norm == pytest.approx(10.0, rel=1e-6)
    assert meta.clipped_norm == pytest.approx(1.0, rel=1e-5)
    assert meta.num_parameters == 25
    total = torch.sqrt(sum((t**2).sum() for t in clipped.values()))
    assert float(total) == pytest.approx(1.0, rel=1e-5)
    # an update inside the ball is left alone
    same, _ = mech._clip_update(small_state(), max_norm=1.0)
    torch.testing.assert_close(same["w"], small_state()["w"])


def test_factory_and_budget_loop():
    assert isinstance(
        PrivacyMechanismFactory.create(PrivacyType.CENTRAL, CFG), CentralPrivacyMechanism
    )
    assert isinstance(
        PrivacyMechanismFactory.create(PrivacyType.LOCAL, CFG), LocalPrivacyMechanism
    )
    with pytest.raises(ValueError, match="Unknown privacy type"):
        PrivacyMechanismFactory.create("federated", CFG)
    acc = RecordingAccountant()
    mech = PrivacyMechanismFactory.create(
        PrivacyType.CENTRAL, CFG, accountant=acc, noise_generator=ones_generator()
    )
    rounds = 0
    while mech.validate_budget():
        mech.add_noise(small_state(), batch_size=4)
        rounds += 1
    assert rounds == 3
    assert mech.get_privacy_spent().epsilon_spent == pytest.approx(0.3)


def test_real_accountant_and_generator_integration():
    conf = PrivacyConfig(epsilon=5.0, delta=1e-5, max_gradient_norm=100.0, noise_multiplier=1.0)
    mech = CentralPrivacyMechanism(
        conf, GaussianAccountant(conf), GaussianNoiseGenerator(seed=3)
    )
    state = {"w": torch.zeros(1000)}
    out = mech.add_noise(state, batch_size=10)
    assert float(out["w"].std()) == pytest.approx(1.0 * 100.0 / 10, rel=0.1)
    assert mech.get_privacy_spent().epsilon_spent > 0
