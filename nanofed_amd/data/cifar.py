"""CIFAR-10 from its *binary* distribution (the dataset BASELINE config 4 names; the
reference ships a loader only for the dataset of its own model,
nanofed/data/mnist.py:9-40 - this is the same entry for the ResNet-18 configuration).

Format (``cifar-10-binary.tar.gz`` -> ``cifar-10-batches-bin/``): ``data_batch_1.bin`` ..
``data_batch_5.bin`` and ``test_batch.bin``, each 10,000 records of 3,073 bytes - one label
byte, then 1,024 red, 1,024 green and 1,024 blue bytes, rows first.  The files are read
with ``numpy.fromfile``; the archive is read member by member without unpacking it.

The "python version" of the dataset (``cifar-10-batches-py``) is a set of pickles.
Unpickling runs code from the file, so that version is refused with a message that names
the binary one; nothing here ever unpickles.

Nothing is downloaded by default (the 170 MB archive is fetched only with
``download=True``); with ``synthetic_fallback=True`` a missing dataset is replaced by the
synthetic CIFAR-shaped set and a warning.
"""

from __future__ import annotations

import tarfile
import urllib.request
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset, TensorDataset

from nanofed_amd.utils import Logger

from .synthetic import synthetic_cifar10_hard

CIFAR10_MEAN = (0.4914, 0.4822, 0.4465)
CIFAR10_STD = (0.2470, 0.2435, 0.2616)
CIFAR10_CLASSES = ("airplane", "automobile", "bird", "cat", "deer", "dog", "frog", "horse",
                   "ship", "truck")

RECORD_BYTES = 1 + 3 * 32 * 32
_DIR = "cifar-10-batches-bin"
_ARCHIVE = "cifar-10-binary.tar.gz"
_URL = "https://www.cs.toronto.edu/~kriz/cifar-10-binary.tar.gz"
_TRAIN = tuple(f"data_batch_{i}.bin" for i in range(1, 6))
_TEST = ("test_batch.bin",)


def parse_records(raw: np.ndarray, source: str = "buffer") -> tuple[np.ndarray, np.ndarray]:
    """``(uint8 [N,3,32,32], int64 [N])`` from the bytes of one batch file."""
    raw = np.asarray(raw, dtype=np.uint8).reshape(-1)
    if raw.size == 0 or raw.size % RECORD_BYTES:
        raise ValueError(
            f"{source}: {raw.size} bytes is not a whole number of {RECORD_BYTES}-byte CIFAR-10 records")
    rec = raw.reshape(-1, RECORD_BYTES)
    labels = rec[:, 0].astype(np.int64)
    if labels.max(initial=0) > 9:
        raise ValueError(f"{source}: label {int(labels.max())} is not a CIFAR-10 class (is this CIFAR-100?)")
    return rec[:, 1:].reshape(-1, 3, 32, 32), labels


def _from_directory(folder: Path, names: tuple[str, ...]):
    if not all((folder / n).exists() for n in names):
        return None
    parts = [parse_records(np.fromfile(folder / n, dtype=np.uint8), str(folder / n)) for n in names]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def _from_archive(archive: Path, names: tuple[str, ...]):
    """Read the wanted members straight out of the tar file (nothing is unpacked: member
    names are only compared, never used as paths)."""
    found: dict[str, tuple[np.ndarray, np.ndarray]] = {}
    with tarfile.open(archive, "r:*") as tar:
        for member in tar:
            base = member.name.rsplit("/", 1)[-1]
            if not member.isfile() or base not in names or base in found:
                continue
            handle = tar.extractfile(member)
            if handle is None:
                continue
            found[base] = parse_records(np.frombuffer(handle.read(), dtype=np.uint8),
                                        f"{archive}:{member.name}")
    if len(found) != len(names):
        return None
    return (np.concatenate([found[n][0] for n in names]),
            np.concatenate([found[n][1] for n in names]))


def find_cifar10(data_dir: str | Path, train: bool = True):
    """The arrays of a split, from ``<data_dir>/cifar-10-batches-bin`` (or ``<data_dir>``
    itself, or the unpacked-by-nobody archive), or ``None`` when none of them is there."""
    root = Path(data_dir)
    names = _TRAIN if train else _TEST
    for folder in (root / _DIR, root):
        got = _from_directory(folder, names)
        if got is not None:
            return got
    for archive in (root / _ARCHIVE, root / _DIR / _ARCHIVE):
        if archive.exists():
            got = _from_archive(archive, names)
            if got is not None:
                return got
    return None


def load_cifar10_arrays(
    data_dir: str | Path,
    train: bool = True,
    download: bool = False,
    synthetic_fallback: bool = False,
) -> tuple[torch.Tensor, torch.Tensor]:
    """``(uint8 [N,3,32,32], int64 [N])`` of the requested split."""
    root = Path(data_dir)
    got = find_cifar10(root, train)
    if got is None and download:
        try:
            root.mkdir(parents=True, exist_ok=True)
            urllib.request.urlretrieve(_URL, root / _ARCHIVE)
            got = find_cifar10(root, train)
        except Exception:
            got = None
    if got is not None:
        images, labels = got
        return torch.from_numpy(np.ascontiguousarray(images)), torch.from_numpy(labels)
    pickled = root / "cifar-10-batches-py"
    hint = ""
    if pickled.exists() or (root / "cifar-10-python.tar.gz").exists():
        hint = (" Found the pickled 'python version' - it is not read (unpickling executes "
                "code from the file); fetch cifar-10-binary.tar.gz instead.")
    if synthetic_fallback:
        Logger().warning("CIFAR-10 binary batches not found under %s; using synthetic data.%s"
                         % (root, hint))
        return synthetic_cifar10_hard(50000 if train else 10000, seed=0 if train else 1)
    raise RuntimeError(
        f"CIFAR-10 binary batches not found under {root} (looked for {_DIR}/{names_of(train)[0]} "
        f"... and {_ARCHIVE}).{hint} Place them there, pass download=True or "
        "synthetic_fallback=True.")


def names_of(train: bool) -> tuple[str, ...]:
    return _TRAIN if train else _TEST


def normalize_cifar10(images_u8: torch.Tensor) -> torch.Tensor:
    mean = torch.tensor(CIFAR10_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CIFAR10_STD).view(1, 3, 1, 1)
    return (images_u8.float() / 255.0 - mean) / std


class CIFAR10TensorDataset(TensorDataset):
    """``TensorDataset(normalised images, labels)`` that keeps the uint8 images too (a
    device-resident trainer gathers batches from the bytes and normalises in its first
    kernel)."""

    def __init__(self, images_u8: torch.Tensor, labels: torch.Tensor) -> None:
        super().__init__(normalize_cifar10(images_u8), labels)
        self.images_u8 = images_u8
        self.labels_i64 = labels
        self.mean = CIFAR10_MEAN
        self.std = CIFAR10_STD


def load_cifar10_data(
    data_dir: str | Path,
    batch_size: int,
    train: bool = True,
    download: bool = False,
    subset_fraction: float = 1.0,
    synthetic_fallback: bool = False,
    seed: int | None = None,
) -> DataLoader:
    """A ``DataLoader`` with the signature of ``load_mnist_data``."""
    images, labels = load_cifar10_arrays(data_dir, train=train, download=download,
                                         synthetic_fallback=synthetic_fallback)
    dataset: TensorDataset | Subset = CIFAR10TensorDataset(images, labels)
    if subset_fraction < 1.0:
        n = int(len(dataset) * subset_fraction)
        rng = np.random.default_rng(seed)
        chosen = rng.choice(len(dataset), size=n, replace=False)
        dataset = Subset(dataset, [int(i) for i in chosen])
    return DataLoader(dataset, batch_size=batch_size, shuffle=train)


def write_cifar10_batch(path: str | Path, images_u8: torch.Tensor | np.ndarray,
                        labels: torch.Tensor | np.ndarray) -> None:
    """Write ``images`` / ``labels`` as one binary batch file (the inverse of
    ``parse_records``: exports of other data in the format, and the tests' fixtures)."""
    img = np.asarray(images_u8, dtype=np.uint8).reshape(-1, 3 * 32 * 32)
    lab = np.asarray(labels, dtype=np.uint8).reshape(-1, 1)
    if len(img) != len(lab):
        raise ValueError("one label per image is required")
    np.concatenate([lab, img], axis=1).tofile(str(path))
