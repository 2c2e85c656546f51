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


class _FakeSession:
    """Stands in for NativeEpochSession (which needs a GPU): remembers what it was built from."""

    built = 0

    def __init__(self, dataloader, device, lr, wd):
        type(self).built += 1
        self.lr, self.wd = lr, wd
        self.first_label = int(dataloader.dataset.labels_i64[0]) if hasattr(
            dataloader.dataset, "labels_i64") else None


def test_a_session_belongs_to_its_loader_object_and_dies_with_it(monkeypatch):
    """The cache is keyed on the loader itself (weakly), not on its address: a second loader
    over other data never gets the first one's session, a loader that is gone takes its
    session (HBM shard, graphs) with it, and a new learning rate builds a new session."""
    import gc

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    trainer = TorchTrainer(TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1, device="cuda"))
    model = MNISTModel()
    sgd = torch.optim.SGD(model.parameters(), lr=0.1)
    _FakeSession.built = 0
    a, b = _loader(n=128), _loader(n=192)
    sa = native.native_session_for(trainer, a, sgd, factory=_FakeSession)
    assert native.native_session_for(trainer, a, sgd, factory=_FakeSession) is sa
    sb = native.native_session_for(trainer, b, sgd, factory=_FakeSession)
    assert sb is not sa and _FakeSession.built == 2
    cache = trainer.__dict__["_native_sessions"]
    assert len(cache) == 2
    # the loader dies: so does its session, whatever object gets that address next
    del a, sa
    gc.collect()
    assert len(cache) == 1
    for _ in range(8):  # new loaders, some of which may land on the freed address
        fresh = _loader(n=64)
        s = native.native_session_for(trainer, fresh, sgd, factory=_FakeSession)
        assert s is not sb and len(cache) == 2
        del fresh, s
        gc.collect()
        assert len(cache) == 1
    # the dataset behind a loader is replaced: the session is rebuilt
    b.__dict__["dataset"] = _loader(n=320).dataset  # (DataLoader guards normal assignment)
    sb2 = native.native_session_for(trainer, b, sgd, factory=_FakeSession)
    assert sb2 is not sb
    # a new learning rate is an argument of the captured kernels
    sgd.param_groups[0]["lr"] = 0.05
    assert native.native_session_for(trainer, b, sgd, factory=_FakeSession) is not sb2


def test_native_argument_of_the_trainer(monkeypatch):
    import pytest

    from nanofed_amd.core.exceptions import DeviceError

    cfg = TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1, device="cpu")
    model = MNISTModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    with pytest.raises(ValueError):
        TorchTrainer(cfg, native="yes")
    forced = TorchTrainer(cfg, native=True)
    with pytest.raises(DeviceError, match="not a HIP device"):
        forced.train_epoch(model, _loader(), opt, 0)
    off = TorchTrainer(cfg, native=False)
    m = off.train_epoch(model, _loader(), opt, 0)
    assert m.backend == "torch" and off.last_backend == "torch"
    assert "backend" not in m.to_dict()  # the wire format keeps the reference's three fields
    # "auto" on a device: the reason why a call does not qualify is available
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    auto = TorchTrainer(TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1, device="cuda"))
    adam = torch.optim.Adam(model.parameters(), lr=0.1)
    assert "plain SGD" in native.native_epoch_unsupported_reason(auto, model, _loader(), adam)
    assert native.native_epoch_unsupported_reason(auto, model, _loader(), opt) is None
