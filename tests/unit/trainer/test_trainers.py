"""Local training loop, classification trainer, DP trainer, metrics callback
(nanofed/trainer/base.py, torch.py, private.py, callback.py)."""

import json
import math

import pytest
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from nanofed_amd.privacy import PrivacyConfig
from nanofed_amd.privacy.accountant import GaussianAccountant
from nanofed_amd.trainer import (
    BaseTrainer,
    MetricsLogger,
    PrivateTrainer,
    TorchTrainer,
    TrainingConfig,
    TrainingMetrics,
)


def loader(n=40, batch=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 10, generator=g)
    y = (x[:, 0] > 0).long()
    return DataLoader(TensorDataset(x, y), batch_size=batch)


def net():
    torch.manual_seed(1)
    return nn.Sequential(nn.Linear(10, 16), nn.ReLU(), nn.Linear(16, 2))


class Recorder:
    def __init__(self):
        self.starts, self.batches, self.ends = [], [], []

    def on_eopch_start(self, epoch):
        self.starts.append(epoch)

    def on_batch_end(self, batch, metrics):
        self.batches.append((batch, metrics))

    def on_epoch_end(self, epoch, metrics):
        self.ends.append((epoch, metrics))


def test_config_and_metrics_records():
    cfg = TrainingConfig(epochs=2, batch_size=8, learning_rate=0.1)
    assert (cfg.device, cfg.max_batches, cfg.log_interval) == ("cpu", None, 10)
    with pytest.raises(Exception):
        cfg.epochs = 3
    m = TrainingMetrics(loss=0.5, accuracy=0.75, epoch=1, batch=4, samples_processed=40)
    assert m.to_dict() == {"loss": 0.5, "accuracy": 0.75, "samples_processed": 40}
    with pytest.raises(TypeError):
        BaseTrainer(cfg)  # abstract


def test_train_epoch_steps_the_optimizer_and_returns_epoch_means():
    cfg = TrainingConfig(epochs=1, batch_size=8, learning_rate=0.1)
    rec = Recorder()
    trainer = TorchTrainer(cfg, callbacks=[rec])
    model = net()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    steps = []
    real_step = opt.step
    opt.step = lambda *a, **k: (steps.append(1), real_step(*a, **k))[1]
    before = [p.clone() for p in model.parameters()]
    out = trainer.train_epoch(model, loader(), opt, epoch=3)
    assert len(steps) == 5 and rec.starts == [3] and len(rec.batches) == 5
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    assert out.loss > 0 and 0.0 <= out.accuracy <= 1.0
    assert out.samples_processed == 40 and out.epoch == 3
    # the returned metrics are the epoch aggregate handed to the callbacks
    assert rec.ends[0][1] == out
    assert out.loss == pytest.approx(sum(m.loss for _, m in rec.batches) / 5, rel=1e-5)
    assert rec.batches[-1][1].samples_processed == 40


def test_max_batches_limits_the_epoch_and_the_mean():
    cfg = TrainingConfig(epochs=1, batch_size=8, learning_rate=0.1, max_batches=2)
    rec = Recorder()
    trainer = TorchTrainer(cfg, callbacks=[rec])
    model = net()
    out = trainer.train_epoch(model, loader(), torch.optim.SGD(model.parameters(), lr=0.1), 0)
    assert len(rec.batches) == 2 and out.samples_processed == 16
    assert out.loss == pytest.approx(sum(m.loss for _, m in rec.batches) / 2, rel=1e-5)


def test_loss_and_accuracy_on_literal_logits():
    trainer = TorchTrainer(TrainingConfig(epochs=1, batch_size=2, learning_rate=0.1))
    logits = torch.tensor([[2.0, 0.0], [0.0, 3.0], [1.0, 0.0], [0.0, 1.0]])
    target = torch.tensor([0, 1, 1, 1])
    want = torch.nn.functional.cross_entropy(logits, target)
    assert float(trainer.compute_loss(logits, target)) == pytest.approx(float(want))
    assert trainer.compute_accuracy(logits, target) == pytest.approx(0.75)
    assert isinstance(trainer.compute_accuracy(logits, target), float)


def test_validate_reports_held_out_loss_and_accuracy():
    trainer = TorchTrainer(TrainingConfig(epochs=1, batch_size=8, learning_rate=0.1))
    model = net()
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    for e in range(20):
        trainer.train_epoch(model, loader(n=200, batch=20), opt, e)
    res = trainer.validate(model, loader(n=100, batch=25, seed=5))
    assert res["samples"] == 100 and res["accuracy"] > 0.8 and res["loss"] < 0.6


def test_metrics_logger_writes_json(tmp_path):
    cb = MetricsLogger(tmp_path / "logs", "exp")
    cb.on_eopch_start(0)
    cb.on_batch_end(0, TrainingMetrics(0.9, 0.5, 0, 0, 8))
    cb.on_batch_end(1, TrainingMetrics(0.7, 0.6, 0, 1, 16))
    cb.on_epoch_end(0, TrainingMetrics(0.8, 0.55, 0, 1, 16))
    files = list((tmp_path / "logs").glob("exp_*.json"))
    assert len(files) == 1
    rows = json.loads(files[0].read_text())
    assert [r["type"] for r in rows] == ["batch", "batch", "epoch"]
    assert rows[1]["batch"] == 1 and rows[2]["loss"] == 0.8 and "timestamp" in rows[0]


# -- DP trainer ----------------------------------------------------------------
PCFG = PrivacyConfig(epsilon=10.0, delta=1e-5, max_gradient_norm=1.0, noise_multiplier=1.0)
TCFG = TrainingConfig(epochs=1, batch_size=8, learning_rate=0.05)


def batch(n=8):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, 10, generator=g) * 5
    return x, (x[:, 0] > 0).long()


def test_clipping_bounds_the_global_gradient_norm():
    trainer = PrivateTrainer(TCFG, PCFG)
    model = net()
    x, y = batch()
    trainer.compute_loss(model(x) * 50, y).backward()
    raw = math.sqrt(sum(float(p.grad.pow(2).sum()) for p in model.parameters()))
    assert raw > 1.0
    trainer._clip_gradients(model)
    clipped = math.sqrt(sum(float(p.grad.pow(2).sum()) for p in model.parameters()))
    assert clipped == pytest.approx(1.0, rel=1e-4)


def test_noise_changes_gradients_and_is_accounted():
    trainer = PrivateTrainer(TCFG, PCFG)
    model = net()
    x, y = batch()
    trainer.compute_loss(model(x), y).backward()
    before = [p.grad.clone() for p in model.parameters()]
    trainer._apply_privacy_to_gradients(model, batch_size=8)
    assert all(not torch.equal(a, p.grad) for a, p in zip(before, model.parameters()))
    assert trainer.get_privacy_spent().epsilon_spent > 0


@pytest.mark.parametrize("mode", ["batch", "per_sample"])
def test_train_batch_metrics_and_budget_tracking(mode):
    trainer = PrivateTrainer(TCFG, PCFG, clip_mode=mode)
    model = net()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    eps = []
    for _ in range(5):
        m = trainer.train_batch(model, batch(), opt)
        assert m.samples_processed == 8 and m.epoch == 0 and m.batch == 0
        assert m.loss > 0 and 0 <= m.accuracy <= 1
        eps.append(trainer.get_privacy_spent().epsilon_spent)
    assert eps == sorted(eps) and eps[0] > 0
    assert trainer._accountant._event_count == 5


def test_per_sample_clipping_matches_a_naive_loop():
    cfg = PrivacyConfig(epsilon=10.0, delta=1e-5, max_gradient_norm=0.5, noise_multiplier=1.0)
    trainer = PrivateTrainer(TCFG, cfg, clip_mode="per_sample")
    model = net()
    x, y = batch(6)
    trainer._per_sample_backward(model, x, y)
    got = [p.grad.clone() for p in model.parameters()]
    want = [torch.zeros_like(p) for p in model.parameters()]
    for i in range(6):
        model.zero_grad()
        trainer.compute_loss(model(x[i : i + 1]), y[i : i + 1]).backward()
        norm = math.sqrt(sum(float(p.grad.pow(2).sum()) for p in model.parameters()))
        coef = min(1.0, 0.5 / (norm + 1e-6))
        for w, p in zip(want, model.parameters()):
            w += p.grad * coef
    for g, w in zip(got, want):
        torch.testing.assert_close(g, w, rtol=1e-4, atol=1e-6)


def test_budget_exhaustion_and_injected_collaborators():
    tight = PrivacyConfig(epsilon=0.1, delta=1e-5, max_gradient_norm=1.0, noise_multiplier=1.0)
    acc = GaussianAccountant(tight)
    trainer = PrivateTrainer(TCFG, tight, accountant=acc, callbacks=[Recorder()])
    assert trainer._accountant is acc and len(trainer._callbacks) == 1
    assert trainer.validate_privacy_budget()
    model = net()
    trainer.train_batch(model, batch(), torch.optim.SGD(model.parameters(), lr=0.05))
    assert not trainer.validate_privacy_budget()
    with pytest.raises(ValueError):
        PrivateTrainer(TCFG, tight, clip_mode="nope")
