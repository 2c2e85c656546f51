"""MNIST data loading (reference: nanofed/data/mnist.py:9-40).

``load_mnist_data`` keeps the reference's signature and preprocessing
(``ToTensor`` + ``Normalize((0.1307,), (0.3081,))``, a random subset of
``subset_fraction`` of the split, shuffling for the training split) but reads
the IDX files itself instead of going through torchvision, which is not a
dependency here.  Files are looked up where torchvision would have put them
(``<data_dir>/MNIST/raw``).  When they are absent and cannot be downloaded,
``synthetic_fallback=True`` substitutes the synthetic MNIST-shaped set.

The tensors are normalised once, up front, and served from memory by a
``TensorDataset``; there are no worker processes to fork.
"""

from __future__ import annotations

import gzip
import struct
import urllib.request
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset, TensorDataset

from nanofed_amd.utils import Logger

from .synthetic import synthetic_mnist

MNIST_MEAN = 0.1307
MNIST_STD = 0.3081

_FILES = {
    True: ("train-images-idx3-ubyte", "train-labels-idx1-ubyte"),
    False: ("t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte"),
}
_MIRRORS = (
    "https://ossci-datasets.s3.amazonaws.com/mnist/",
    "http://yann.lecun.com/exdb/mnist/",
)


def read_idx(path: Path) -> np.ndarray:
    """Parse an IDX file (optionally gzip-compressed) into a uint8 array."""
    opener = gzip.open if path.suffix == ".gz" else open
    with opener(path, "rb") as f:
        raw = f.read()
    zero, dtype_code, ndim = struct.unpack(">HBB", raw[:4])
    if zero != 0 or dtype_code != 0x08:
        raise ValueError(f"{path} is not an unsigned-byte IDX file")
    dims = struct.unpack(">" + "I" * ndim, raw[4 : 4 + 4 * ndim])
    data = np.frombuffer(raw, dtype=np.uint8, offset=4 + 4 * ndim)
    return data.reshape(dims)


def _find(raw_dir: Path, stem: str) -> Path | None:
    for name in (stem, stem + ".gz"):
        if (raw_dir / name).exists():
            return raw_dir / name
    return None


def _download(raw_dir: Path, stem: str) -> Path | None:
    raw_dir.mkdir(parents=True, exist_ok=True)
    target = raw_dir / (stem + ".gz")
    for mirror in _MIRRORS:
        try:
            urllib.request.urlretrieve(mirror + stem + ".gz", target)
            return target
        except Exception:
            continue
    return None


def load_mnist_arrays(
    data_dir: str | Path,
    train: bool = True,
    download: bool = True,
    synthetic_fallback: bool = False,
) -> tuple[torch.Tensor, torch.Tensor]:
    """``(uint8 [N,1,28,28], int64 [N])`` for the requested split."""
    raw_dir = Path(data_dir) / "MNIST" / "raw"
    paths = []
    for stem in _FILES[train]:
        p = _find(raw_dir, stem)
        if p is None and download:
            p = _download(raw_dir, stem)
        paths.append(p)
    if all(p is not None for p in paths):
        images = read_idx(paths[0])[:, None, :, :]
        labels = read_idx(paths[1]).astype(np.int64)
        return torch.from_numpy(images.copy()), torch.from_numpy(labels)
    if synthetic_fallback:
        Logger().warning(
            "MNIST files not found and not downloadable; using synthetic data"
        )
        return synthetic_mnist(60000 if train else 10000, seed=0 if train else 1)
    raise RuntimeError(
        f"MNIST files not found under {raw_dir} and download failed. "
        "Place the IDX files there or pass synthetic_fallback=True."
    )


def normalize_mnist(images_u8: torch.Tensor) -> torch.Tensor:
    return (images_u8.float() / 255.0 - MNIST_MEAN) / MNIST_STD


class MNISTTensorDataset(TensorDataset):
    """``TensorDataset(normalised images, labels)`` that also keeps the raw uint8
    images: a trainer that runs the hand-written kernels gathers batches from the
    uint8 rows on the device and normalises inside its first kernel, so it needs
    the bytes, not the floats the ``DataLoader`` would hand out."""

    def __init__(self, images_u8: torch.Tensor, labels: torch.Tensor) -> None:
        super().__init__(normalize_mnist(images_u8), labels)
        self.images_u8 = images_u8
        self.labels_i64 = labels
        self.mean = MNIST_MEAN
        self.std = MNIST_STD


def load_mnist_data(
    data_dir: str | Path,
    batch_size: int,
    train: bool = True,
    download: bool = True,
    subset_fraction: float = 0.2,
    synthetic_fallback: bool = False,
    seed: int | None = None,
) -> DataLoader:
    images, labels = load_mnist_arrays(
        data_dir, train=train, download=download,
        synthetic_fallback=synthetic_fallback,
    )
    dataset: TensorDataset | Subset = MNISTTensorDataset(images, labels)
    if subset_fraction < 1.0:
        n = int(len(dataset) * subset_fraction)
        rng = np.random.default_rng(seed)
        chosen = rng.choice(len(dataset), size=n, replace=False)
        dataset = Subset(dataset, [int(i) for i in chosen])
    return DataLoader(dataset, batch_size=batch_size, shuffle=train)
