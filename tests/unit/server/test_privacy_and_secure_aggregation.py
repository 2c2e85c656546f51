"""Privacy-aware aggregation, threshold sums and the encrypted aggregators
(nanofed/server/aggregator/privacy.py, secure.py)."""

import pytest
import torch
import torch.nn as nn

from nanofed_amd.core import ModelUpdate
from nanofed_amd.privacy.accountant import PrivacySpent
from nanofed_amd.privacy.mechanisms import PrivacyType
from nanofed_amd.server.aggregator import (
    HomomorphicSecureAggregator,
    PrivacyAwareAggregationConfig,
    PrivacyAwareAggregator,
    SecureAggregationConfig,
    SecureAggregationType,
    SecureMaskingAggregator,
    ThresholdSecureAggregation,
)
from nanofed_amd.server.aggregator.secure import PairwiseMaskSecureAggregation
from nanofed_amd.utils import get_current_time


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(10, 5)
        self.fc2 = nn.Linear(5, 2)

    def forward(self, x):
        return self.fc2(self.fc1(x))


def updates(n, samples=None, eps=None, rnd=0):
    torch.manual_seed(n)
    out = []
    for i in range(n):
        u = ModelUpdate(
            client_id=f"c{i}", round_number=rnd,
            model_state={k: v.clone() + i for k, v in Net().state_dict().items()},
            metrics={"loss": 0.5 - 0.1 * i, "accuracy": 0.7 + 0.1 * i,
                     "samples_processed": (samples or [100] * n)[i]},
            timestamp=get_current_time(),
        )
        if eps is not None:
            u["privacy_spent"] = PrivacySpent(eps[i], 1e-5)
        out.append(u)
    return out


def config(**kw):
    base = dict(epsilon=5.0, delta=1e-5, max_gradient_norm=10.0, noise_multiplier=0.5,
                min_clients=2)
    base.update(kw)
    return PrivacyAwareAggregationConfig(**base)


def test_config_defaults_and_bounds():
    cfg = PrivacyAwareAggregationConfig()
    assert cfg.privacy_type is PrivacyType.CENTRAL
    assert cfg.secure_aggregation is SecureAggregationType.NONE
    assert cfg.min_clients >= 1 and 0.0 <= cfg.dropout_tolerance <= 1.0
    with pytest.raises(Exception):
        PrivacyAwareAggregationConfig(min_clients=0)
    with pytest.raises(Exception):
        PrivacyAwareAggregationConfig(dropout_tolerance=1.5)


def test_central_weights_are_sample_proportions():
    agg = PrivacyAwareAggregator(config())
    w = agg._compute_weights(updates(3, samples=[100, 200, 700]))
    assert w == pytest.approx([0.1, 0.2, 0.7])


def test_local_weights_grow_with_the_reported_epsilon():
    agg = PrivacyAwareAggregator(config(privacy_type=PrivacyType.LOCAL))
    w = agg._compute_weights(updates(3, eps=[0.5, 1.0, 2.0]))
    assert w[0] < w[1] < w[2] and sum(w) == pytest.approx(1.0)


@pytest.mark.parametrize("ptype", [PrivacyType.CENTRAL, PrivacyType.LOCAL])
def test_aggregate_produces_a_model_and_privacy_metrics(ptype):
    agg = PrivacyAwareAggregator(config(privacy_type=ptype))
    model = Net()
    ups = updates(3, samples=[100, 100, 200], eps=[1.0, 1.0, 1.0])
    result = agg.aggregate(model, ups)
    assert result.num_clients == 3 and result.round_number == 1
    assert {"privacy_epsilon", "privacy_delta", "accuracy", "loss"} <= set(result.metrics)
    want_acc = 0.7 * 0.25 + 0.8 * 0.25 + 0.9 * 0.5
    assert result.metrics["accuracy"] == pytest.approx(want_acc)
    for p in model.parameters():
        assert torch.isfinite(p).all()
    if ptype is PrivacyType.CENTRAL:
        assert result.metrics["privacy_epsilon"] > 0


def test_threshold_secure_aggregation_sums_the_noised_states():
    agg = PrivacyAwareAggregator(
        config(secure_aggregation=SecureAggregationType.THRESHOLD, noise_multiplier=1e-6,
               max_gradient_norm=100.0)
    )
    model = Net()
    ups = updates(2)
    agg.aggregate(model, ups)
    want = ups[0]["model_state"]["fc2.bias"] + ups[1]["model_state"]["fc2.bias"]
    torch.testing.assert_close(model.fc2.bias.data, want, rtol=1e-4, atol=1e-4)


def test_validation_messages():
    agg = PrivacyAwareAggregator(config(min_clients=3))
    with pytest.raises(ValueError, match="Not enough clients"):
        agg.aggregate(Net(), updates(2))
    agg = PrivacyAwareAggregator(config(privacy_type=PrivacyType.LOCAL))
    with pytest.raises(ValueError, match="Missing privacy budget"):
        agg.aggregate(Net(), updates(2))
    agg = PrivacyAwareAggregator(config())
    bad = updates(2)
    bad[1]["model_state"] = {"other": torch.zeros(1)}
    with pytest.raises(ValueError, match="Inconsistent model architectures"):
        agg.aggregate(Net(), bad)
    mixed = updates(2)
    mixed[1]["round_number"] = 4
    with pytest.raises(ValueError, match="Updates from different rounds"):
        agg.aggregate(Net(), mixed)


def test_threshold_share_sum_and_verification():
    t = ThresholdSecureAggregation(min_clients=2)
    shares = [{"w": torch.ones(2, 2) * (i + 1)} for i in range(3)]
    assert t.verify_shares(shares)
    torch.testing.assert_close(t.aggregate_shares(shares)["w"], torch.full((2, 2), 6.0))
    assert not t.verify_shares(shares[:1])
    assert not t.verify_shares([{"w": torch.ones(2, 2)}, {"w": torch.ones(3)}])
    with pytest.raises(ValueError, match="Not enough clients"):
        t.aggregate_shares(shares[:1])


# -- encrypted aggregators ---------------------------------------------------
SEC = SecureAggregationConfig(min_clients=2, key_size=2048)


def test_rsa_round_trip_chunks_and_tamper_detection():
    agg = HomomorphicSecureAggregator(SEC)
    upd = {"small": torch.tensor([1.0, -2.0, 3.5]), "big": torch.randn(100, 100)}
    enc = agg.encrypt_update(upd)
    chunk = 2048 // 8 - 66  # OAEP-SHA256 payload of a 2048-bit key
    assert len(enc["big"]) == -(-100 * 100 * 4 // chunk)
    assert all(len(c) == 256 for c in enc["big"])
    dec = agg.decrypt_aggregate(enc)
    assert torch.equal(dec["small"], upd["small"]) and torch.equal(dec["big"], upd["big"])
    assert agg.encrypt_update({}) == {}
    enc["small"][0] = bytes(256)
    with pytest.raises(ValueError, match="Decryption failed for small"):
        agg.decrypt_aggregate(enc)


def test_rsa_aggregate_of_two_clients_is_their_sum():
    agg = HomomorphicSecureAggregator(SEC)
    a, b = {"w": torch.tensor([1.0, 2.0])}, {"w": torch.tensor([10.0, 20.0])}
    total = agg.decrypt_aggregate(
        agg.aggregate_encrypted([agg.encrypt_update(a), agg.encrypt_update(b)])
    )
    torch.testing.assert_close(total["w"], torch.tensor([11.0, 22.0]))
    with pytest.raises(ValueError, match="Need at least 2 clients"):
        agg.aggregate_encrypted([agg.encrypt_update(a)])


def test_masking_aggregator_sum_tamper_and_consecutive_rounds():
    agg = SecureMaskingAggregator(SEC, key=bytes(range(32)))
    for rnd in range(2):
        a = {"w": torch.randn(4, 3), "b": torch.randn(3)}
        b = {"w": torch.randn(4, 3), "b": torch.randn(3)}
        ea, eb = agg.encrypt_update(a), agg.encrypt_update(b)
        assert isinstance(ea["w"], bytes)
        total = agg.decrypt_aggregate(agg.aggregate_encrypted([ea, eb]))
        torch.testing.assert_close(total["w"], a["w"] + b["w"], atol=1e-4, rtol=0)
        torch.testing.assert_close(total["b"], a["b"] + b["b"], atol=1e-4, rtol=0)
    with pytest.raises(ValueError, match="Need at least 2 clients"):
        agg.aggregate_encrypted([agg.encrypt_update(a)])
    enc = agg.encrypt_update(a)
    enc["w"] = enc["w"][:-1] + bytes([enc["w"][-1] ^ 1])
    with pytest.raises(ValueError, match="Decryption failed for w"):
        agg.decrypt_aggregate(enc)
    with pytest.raises(ValueError):
        SecureMaskingAggregator(SEC, key=b"short")
    SecureMaskingAggregator(SEC)  # derives its own key


def test_pairwise_masks_cancel_and_dropouts_are_repaired():
    n = 5
    proto = PairwiseMaskSecureAggregation(num_clients=n, mask_scale=8.0, secret=b"s" * 32)
    flats = [torch.randn(1000) for _ in range(n)]
    masked = [proto.mask(i, f) for i, f in enumerate(flats)]
    # a single blinded buffer reveals nothing usable
    assert float((masked[0] - flats[0]).abs().mean()) > 1.0
    total = proto.aggregate_flat(masked)
    torch.testing.assert_close(total, sum(flats), atol=1e-3, rtol=0)
    present = [0, 2, 3]
    partial = proto.aggregate_flat([masked[i] for i in present], present=present)
    torch.testing.assert_close(partial, sum(flats[i] for i in present), atol=1e-3, rtol=0)
