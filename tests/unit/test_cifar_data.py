"""The CIFAR-10 reader on fixture files written in the binary format (the dataset itself
is not on the boxes: parity with real CIFAR-10 is unpinned), the refusal of the pickled
version, and the data-shape test that admits an image set to the grouped ResNet kernels."""

import io
import tarfile

import numpy as np
import pytest
import torch

from nanofed_amd.data import cifar
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.engine.data import build_federated_data
from nanofed_amd.engine.resnet_native import cifar_shaped
from nanofed_amd.parallel import Communicator


def _fixture(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, size=(n, 3, 32, 32), dtype=np.uint8),
            rng.integers(0, 10, size=n).astype(np.int64))


def _write_split(folder, per_file=8):
    folder.mkdir(parents=True, exist_ok=True)
    train = []
    for i, name in enumerate(cifar.names_of(True)):
        x, y = _fixture(per_file, 10 + i)
        cifar.write_cifar10_batch(folder / name, x, y)
        train.append((x, y))
    tx, ty = _fixture(per_file, 99)
    cifar.write_cifar10_batch(folder / "test_batch.bin", tx, ty)
    return (np.concatenate([t[0] for t in train]), np.concatenate([t[1] for t in train])), (tx, ty)


def test_record_layout_label_then_red_green_blue_planes():
    x, y = _fixture(3, 0)
    raw = np.concatenate([y.astype(np.uint8)[:, None], x.reshape(3, -1)], axis=1).reshape(-1)
    assert raw.size == 3 * 3073
    img, lab = cifar.parse_records(raw)
    assert np.array_equal(img, x) and np.array_equal(lab, y)
    # byte 1 of a record is the red value of pixel (0, 0), byte 1025 the green one
    assert raw[1] == x[0, 0, 0, 0] and raw[1025] == x[0, 1, 0, 0] and raw[2049] == x[0, 2, 0, 0]
    with pytest.raises(ValueError, match="whole number"):
        cifar.parse_records(raw[:-1])
    bad = raw.copy()
    bad[0] = 57
    with pytest.raises(ValueError, match="not a CIFAR-10 class"):
        cifar.parse_records(bad)


def test_reads_the_directory_and_the_archive(tmp_path):
    (train, test) = _write_split(tmp_path / "a" / "cifar-10-batches-bin")
    x, y = cifar.load_cifar10_arrays(tmp_path / "a", train=True)
    assert x.dtype == torch.uint8 and tuple(x.shape) == (40, 3, 32, 32)
    assert np.array_equal(x.numpy(), train[0]) and np.array_equal(y.numpy(), train[1])
    ex, ey = cifar.load_cifar10_arrays(tmp_path / "a", train=False)
    assert np.array_equal(ex.numpy(), test[0]) and np.array_equal(ey.numpy(), test[1])

    # the same files inside cifar-10-binary.tar.gz, read without unpacking
    (tmp_path / "b").mkdir()
    with tarfile.open(tmp_path / "b" / "cifar-10-binary.tar.gz", "w:gz") as tar:
        for f in sorted((tmp_path / "a" / "cifar-10-batches-bin").iterdir()):
            tar.add(f, arcname=f"cifar-10-batches-bin/{f.name}")
        junk = tarfile.TarInfo("cifar-10-batches-bin/../../escape.bin")
        junk.size = 4
        tar.addfile(junk, io.BytesIO(b"evil"))
    x2, y2 = cifar.load_cifar10_arrays(tmp_path / "b", train=True)
    assert torch.equal(x2, x) and torch.equal(y2, y)
    assert not (tmp_path / "escape.bin").exists() and sorted(p.name for p in (tmp_path / "b").iterdir()) == [
        "cifar-10-binary.tar.gz"]


def test_pickled_version_is_refused_not_loaded(tmp_path):
    (tmp_path / "cifar-10-batches-py").mkdir()
    (tmp_path / "cifar-10-batches-py" / "data_batch_1").write_bytes(b"\x80\x04N.")
    with pytest.raises(RuntimeError, match="pickled 'python version'"):
        cifar.load_cifar10_arrays(tmp_path, train=True)
    # with the fallback: synthetic data of the CIFAR shape
    x, y = cifar.load_cifar10_arrays(tmp_path, train=False, synthetic_fallback=True)
    assert tuple(x.shape) == (10000, 3, 32, 32) and int(y.max()) <= 9


def test_loader_normalises_per_channel(tmp_path):
    _write_split(tmp_path / "cifar-10-batches-bin")
    loader = cifar.load_cifar10_data(tmp_path, batch_size=10, train=False)
    xb, yb = next(iter(loader))
    raw = loader.dataset.images_u8[:8].float() / 255.0
    want = (raw - torch.tensor(cifar.CIFAR10_MEAN).view(1, 3, 1, 1)) / torch.tensor(
        cifar.CIFAR10_STD).view(1, 3, 1, 1)
    torch.testing.assert_close(xb, want)
    assert len(loader.dataset) == 8 and yb.dtype == torch.int64
    half = cifar.load_cifar10_data(tmp_path, batch_size=4, train=True, subset_fraction=0.5, seed=1)
    assert len(half.dataset) == 20


def test_engine_trains_on_files_and_the_shape_test_admits_them(tmp_path):
    _write_split(tmp_path / "cifar-10-batches-bin", per_file=16)
    cfg = EngineConfig(model="resnet18", dataset="cifar10", data_dir=str(tmp_path), num_clients=2,
                       samples_per_client=32, batch_size=8, eval_samples=16,
                       model_kwargs={"width": 8}, local_epochs=1)
    data = build_federated_data(cfg)
    assert tuple(data.inputs.shape) == (80, 3, 32, 32) and len(data.eval_inputs) == 16
    assert data.mean == cifar.CIFAR10_MEAN
    assert cifar_shaped(data)
    mnist = build_federated_data(EngineConfig(num_clients=2, samples_per_client=16, eval_samples=8))
    assert not cifar_shaped(mnist) and not cifar_shaped(None)
    eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cpu")))
    assert eng.run_round().status == "COMPLETED"


def test_hard_synthetic_cifar_has_noisy_labels_and_several_patterns():
    from nanofed_amd.data import synthetic_cifar10, synthetic_cifar10_hard

    x, y = synthetic_cifar10_hard(4000, seed=3)
    assert tuple(x.shape) == (4000, 3, 32, 32) and x.dtype == torch.uint8
    # a nearest-class-mean classifier separates the easy set completely, the hard one not
    def centroid_accuracy(make):
        a, la = make(4000, seed=3)
        b, lb = make(1000, seed=4)
        fa, fb = a.float().flatten(1), b.float().flatten(1)
        cents = torch.stack([fa[la == c].mean(0) for c in range(10)])
        pred = torch.cdist(fb, cents).argmin(1)
        return float((pred == lb).float().mean())

    assert centroid_accuracy(synthetic_cifar10) > 0.99
    assert centroid_accuracy(synthetic_cifar10_hard) < 0.93
