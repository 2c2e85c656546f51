"""Closed forms of the accountants (nanofed/privacy/accountant/*.py).

The sampling rate is q = min(samples / max_gradient_norm, 1) - the reference's
definition, pinned here to 1e-9 relative."""

import math

import pytest

from nanofed_amd.privacy import PrivacyConfig
from nanofed_amd.privacy.accountant import (
    GaussianAccountant,
    PrivacySpent,
    RDPAccountant,
)
from nanofed_amd.privacy.accountant.rdp import SubsampledRDPAccountant, subsampled_gaussian_rdp
from nanofed_amd.privacy.exceptions import PrivacyError


def cfg(**kw):
    base = dict(epsilon=10.0, delta=1e-5, max_gradient_norm=1000.0, noise_multiplier=1.0)
    base.update(kw)
    return PrivacyConfig(**base)


def c_of(delta):
    return math.sqrt(2 * math.log(1.25 / delta))


def test_nothing_spent_before_the_first_event():
    for acc in (GaussianAccountant(cfg()), RDPAccountant(cfg())):
        spent = acc.get_privacy_spent()
        if isinstance(acc, GaussianAccountant):
            assert spent == PrivacySpent(0.0, 0.0)
        assert acc.validate_budget()


def test_gaussian_single_event_closed_form():
    acc = GaussianAccountant(cfg())
    acc.add_noise_event(sigma=2.0, samples=100)
    want = c_of(1e-5) * (100 / 1000.0) / 2.0
    spent = acc.get_privacy_spent()
    assert math.isclose(spent.epsilon_spent, want, rel_tol=1e-9)
    assert spent.delta_spent == 1e-5


def test_gaussian_is_linear_in_samples_and_inverse_in_sigma():
    def eps(sigma, samples):
        a = GaussianAccountant(cfg())
        a.add_noise_event(sigma, samples)
        return a.get_privacy_spent().epsilon_spent

    assert math.isclose(eps(1.0, 200), 2 * eps(1.0, 100), rel_tol=1e-9)
    assert math.isclose(eps(2.0, 100), eps(1.0, 100) / 2, rel_tol=1e-9)
    # the rate is capped at 1
    assert math.isclose(eps(1.0, 5000), eps(1.0, 1000), rel_tol=1e-9)
    assert math.isclose(eps(1.0, 1000), c_of(1e-5), rel_tol=1e-9)


def test_gaussian_composition_is_additive():
    acc = GaussianAccountant(cfg())
    events = [(1.0, 100), (2.0, 300), (0.5, 50)]
    for s, n in events:
        acc.add_noise_event(s, n)
    want = sum(c_of(1e-5) * min(n / 1000.0, 1.0) / s for s, n in events)
    assert math.isclose(acc.get_privacy_spent().epsilon_spent, want, rel_tol=1e-9)


@pytest.mark.parametrize("acc_cls", [GaussianAccountant, RDPAccountant])
def test_invalid_event_arguments(acc_cls):
    acc = acc_cls(cfg())
    with pytest.raises(ValueError, match="Number of samples must be positive"):
        acc.add_noise_event(1.0, 0)
    with pytest.raises(ValueError, match="Number of samples must be positive"):
        acc.add_noise_event(1.0, -5)
    with pytest.raises(ValueError, match="Noise multiplier must be positive"):
        acc.add_noise_event(0.0, 10)
    with pytest.raises(ValueError, match="Noise multiplier must be positive"):
        acc.add_noise_event(-1.0, 10)


def test_budget_is_exhausted_when_q_is_one():
    tight = PrivacyConfig(epsilon=0.1, delta=1e-5, max_gradient_norm=1.0, noise_multiplier=1.0)
    acc = GaussianAccountant(tight)
    assert acc.validate_budget()
    acc.add_noise_event(1.0, 32)  # q = min(32 / 1, 1) = 1 -> eps = c >> 0.1
    assert not acc.validate_budget()
    roomy = PrivacyConfig(epsilon=10.0, delta=1e-5)
    assert acc.validate_budget(roomy)


def test_rdp_orders_are_validated():
    with pytest.raises(PrivacyError):
        RDPAccountant(cfg(), orders=[1.0, 2.0])
    with pytest.raises(PrivacyError):
        RDPAccountant(cfg(), orders=[0.5])
    acc = RDPAccountant(cfg())
    assert sorted(acc._rdp_budget) == [1.5, 2.0, 2.5, 3.0, 4.0, 8.0, 16.0, 32.0, 64.0]


def test_rdp_closed_form_and_conversion():
    orders = [2.0, 4.0, 8.0]
    acc = RDPAccountant(cfg(), orders=orders)
    acc.add_noise_event(sigma=1.5, samples=100)
    acc.add_noise_event(sigma=1.5, samples=100)
    q = 0.1
    rdp = {a: 2 * q * q * a / (2 * 1.5**2) for a in orders}
    for a in orders:
        assert math.isclose(acc._rdp_budget[a], rdp[a], rel_tol=1e-9)
    want = min(rdp[a] + math.log(1 / 1e-5) / (a - 1) for a in orders)
    spent = acc.get_privacy_spent()
    assert math.isclose(spent.epsilon_spent, want, rel_tol=1e-9)
    assert spent.delta_spent == 1e-5


def test_rdp_is_monotone_in_events_and_amplified_by_small_q():
    a = RDPAccountant(cfg())
    last = a.get_privacy_spent().epsilon_spent
    for _ in range(10):
        a.add_noise_event(1.0, 500)
        now = a.get_privacy_spent().epsilon_spent
        assert now >= last
        last = now
    small, large = RDPAccountant(cfg()), RDPAccountant(cfg())
    for _ in range(20):
        small.add_noise_event(1.0, 10)
        large.add_noise_event(1.0, 900)
    assert small.get_privacy_spent().epsilon_spent < large.get_privacy_spent().epsilon_spent


def test_compute_rdp_gaussian_helper():
    acc = RDPAccountant(cfg(), orders=[2.0, 3.0])
    inc = acc._compute_rdp_gaussian(sigma=2.0, sampling_rate=0.5)
    assert math.isclose(inc[2.0], 0.25 * 2 / 8, rel_tol=1e-12)
    assert math.isclose(inc[3.0], 0.25 * 3 / 8, rel_tol=1e-12)


@pytest.mark.parametrize("acc_cls, n", [(GaussianAccountant, 5000), (RDPAccountant, 1000)])
def test_many_events_stay_finite(acc_cls, n):
    acc = acc_cls(cfg(max_gradient_norm=1e6))
    for i in range(n):
        acc.add_noise_event(1.0 + (i % 7) * 0.1, 1 + i % 64)
    spent = acc.get_privacy_spent()
    assert math.isfinite(spent.epsilon_spent) and spent.epsilon_spent > 0


def test_privacy_spent_validate_uses_both_components():
    conf = PrivacyConfig(epsilon=1.0, delta=1e-5)
    assert PrivacySpent(1.0, 1e-5).validate(conf)
    assert not PrivacySpent(1.0001, 1e-5).validate(conf)
    assert not PrivacySpent(0.5, 2e-5).validate(conf)


def test_subsampled_accountant_amplifies_and_matches_the_unsampled_limit():
    assert math.isclose(subsampled_gaussian_rdp(1.0, 2.0, 8), 8 / (2 * 4.0), rel_tol=1e-12)
    assert subsampled_gaussian_rdp(0.01, 1.0, 8) < 0.01 * subsampled_gaussian_rdp(1.0, 1.0, 8)
    acc = SubsampledRDPAccountant(cfg(), dataset_size=60000)
    for _ in range(100):
        acc.add_noise_event(1.1, 64)
    eps = acc.get_privacy_spent().epsilon_spent
    assert 0 < eps < 2.0
    with pytest.raises(PrivacyError):
        SubsampledRDPAccountant(cfg(), dataset_size=0)
