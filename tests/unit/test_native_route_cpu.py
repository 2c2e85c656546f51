"""The native route of ``TorchTrainer.train_epoch``: which calls qualify, and that
everything else (every call on a CPU) takes the PyTorch path unchanged."""

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from nanofed_amd.data.mnist import MNISTTensorDataset
from nanofed_amd.data.synthetic import synthetic_mnist
from nanofed_amd.models import MNISTModel
from nanofed_amd.trainer import TorchTrainer, TrainingConfig
from nanofed_amd.trainer import native


def _loader(n=256, batch=64, shuffle=True, subset=None):
    images, labels = synthetic_mnist(n, seed=3)
    ds = MNISTTensorDataset(images, labels)
    if subset is not None:
        ds = Subset(ds, subset)
    return DataLoader(ds, batch_size=batch, shuffle=shuffle)


def test_raw_rows_follow_nested_subsets():
    images, labels = synthetic_mnist(100, seed=1)
    ds = MNISTTensorDataset(images, labels)
    nested = Subset(Subset(ds, [10, 20, 30, 40]), [3, 1])
    base, rows = native._raw_dataset(nested)
    assert base is ds and rows.tolist() == [40, 20]
    assert native._raw_dataset(torch.utils.data.TensorDataset(torch.zeros(4, 1))) is None
    x, y = ds[5]
    assert x.dtype == torch.float32 and x.shape == (1, 28, 28) and int(y) == int(labels[5])


def test_on_a_cpu_every_call_takes_the_pytorch_path():
    trainer = TorchTrainer(TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1, device="cpu"))
    model = MNISTModel()
    loader = _loader()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    assert not native.native_epoch_supported(trainer, model, loader, opt)
    before = [p.detach().clone() for p in model.parameters()]
    m = trainer.train_epoch(model, loader, opt, 0)
    assert m.samples_processed == 256 and np.isfinite(m.loss)
    assert any((a != b).any() for a, b in zip(before, model.parameters()))
    assert "_native_sessions" not in trainer.__dict__


def test_only_the_exact_native_step_qualifies(monkeypatch):
    # pretend a GPU is there: the remaining conditions are host-side facts
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    trainer = TorchTrainer(TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1, device="cuda"))
    model = MNISTModel()
    sgd = torch.optim.SGD(model.parameters(), lr=0.1)
    assert native.native_epoch_supported(trainer, model, _loader(), sgd)
    assert native.native_epoch_supported(trainer, model, _loader(subset=list(range(100))), sgd)
    assert native.native_epoch_supported(trainer, model, _loader(batch=32), sgd)
    assert not native.native_epoch_supported(trainer, model, _loader(n=4096, batch=2048), sgd)
    assert not native.native_epoch_supported(trainer, model, _loader(shuffle=False), sgd)
    momentum = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
    assert not native.native_epoch_supported(trainer, model, _loader(), momentum)
    adam = torch.optim.Adam(model.parameters(), lr=0.1)
    assert not native.native_epoch_supported(trainer, model, _loader(), adam)
    partial = torch.optim.SGD(list(model.parameters())[:2], lr=0.1)
    assert not native.native_epoch_supported(trainer, model, _loader(), partial)

    class Wider(MNISTModel):
        pass

    assert not native.native_epoch_supported(trainer, Wider(), _loader(), sgd)
    plain = DataLoader(torch.utils.data.TensorDataset(torch.zeros(64, 1, 28, 28),
                                                      torch.zeros(64, dtype=torch.long)),
                       batch_size=64, shuffle=True)
    assert not native.native_epoch_supported(trainer, model, plain, sgd)
    capped = TorchTrainer(TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1, device="cuda",
                                         max_batches=3))
    assert not native.native_epoch_supported(capped, model, _loader(), sgd)
    monkeypatch.setenv("NANOFED_NATIVE_TRAINER", "0")
    assert not native.native_epoch_supported(trainer, model, _loader(), sgd)
