"""Core types, utils and data helpers."""

import asyncio
import logging
from datetime import timezone

import numpy as np
import pytest
import torch

import nanofed_amd
from nanofed_amd import core
from nanofed_amd.data import (
    dirichlet_partition,
    iid_partition,
    label_histogram,
    load_mnist_data,
    sized_partition,
    synthetic_cifar10,
    synthetic_mnist,
    synthetic_tokens,
)
from nanofed_amd.data.mnist import read_idx
from nanofed_amd.utils import LogConfig, Logger, LogLevel, get_current_time, log_exec


def test_public_api_surface():
    for name in ["HTTPClient", "HTTPServer", "TrainingConfig", "TorchTrainer", "PrivateTrainer",
                 "Coordinator", "CoordinatorConfig", "FedAvgAggregator", "ModelManager",
                 "coordinate", "__version__"]:
        assert hasattr(nanofed_amd, name), name
    # the reference's public (misspelled) names and their corrected aliases
    assert core.AggregatorProtoocol is not None
    assert getattr(core, "AggregatorProtocol", core.AggregatorProtoocol) is not None
    assert issubclass(core.AggregationError, core.NanoFedError)
    assert issubclass(core.ModelManagerError, core.NanoFedError)


def test_model_version_is_frozen():
    from pathlib import Path

    v = core.ModelVersion("id", get_current_time(), {"a": 1}, Path("/tmp/x"))
    with pytest.raises(Exception):
        v.version_id = "other"


def test_clock_is_timezone_aware_utc():
    now = get_current_time()
    assert now.tzinfo is not None and now.utcoffset().total_seconds() == 0
    assert now.tzinfo == timezone.utc


def test_logger_is_a_singleton_with_contexts_and_log_exec(caplog):
    a, b = Logger(), Logger()
    assert a is b
    a.configure(LogConfig(level=LogLevel.DEBUG))
    with a.context("coordinator", "round_3"):
        assert "coordinator" in str(a.current_context) and "round_3" in str(a.current_context)
    assert a.current_context is None

    @log_exec
    def work(x):
        return x * 2

    @log_exec
    async def awork(x):
        await asyncio.sleep(0)
        return x + 1

    @log_exec
    def boom():
        raise RuntimeError("no")

    with caplog.at_level(logging.DEBUG, logger="nanofed"):
        assert work(4) == 8
        assert asyncio.run(awork(4)) == 5
        with pytest.raises(RuntimeError):
            boom()
    text = caplog.text
    assert "Starting work" in text and "Completed work in" in text
    assert "Completed awork in" in text and "Error in boom" in text
    a.configure(LogConfig(level=LogLevel.ERROR))


def test_partitions_are_disjoint_and_deterministic():
    parts = iid_partition(1000, 8, seed=1)
    allidx = np.concatenate(parts)
    assert len(allidx) == 1000 and len(set(allidx.tolist())) == 1000
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert all(np.array_equal(a, b) for a, b in zip(parts, iid_partition(1000, 8, seed=1)))
    sized = sized_partition(1000, [500, 300, 100], seed=2)
    assert [len(s) for s in sized] == [500, 300, 100]
    assert len(set(np.concatenate(sized).tolist())) == 900
    with pytest.raises(ValueError):
        sized_partition(10, [8, 8])


def test_dirichlet_partition_skews_labels():
    labels = np.random.default_rng(0).integers(0, 10, size=20000)
    skew = dirichlet_partition(labels, 8, alpha=0.1, seed=0, min_size=10)
    flat = dirichlet_partition(labels, 8, alpha=100.0, seed=0, min_size=10)
    assert len(set(np.concatenate(skew).tolist())) == 20000
    h_skew = label_histogram(labels, skew, 10).astype(float)
    h_flat = label_histogram(labels, flat, 10).astype(float)
    top_share = lambda h: (h.max(1) / h.sum(1)).mean()
    assert top_share(h_skew) > 0.4 > top_share(h_flat)
    with pytest.raises(ValueError):
        dirichlet_partition(labels, 8, alpha=0.0)


def test_synthetic_datasets_have_the_right_shapes_and_are_learnable():
    x, y = synthetic_mnist(256, seed=0)
    assert x.shape == (256, 1, 28, 28) and x.dtype == torch.uint8 and y.dtype == torch.int64
    x2, _ = synthetic_mnist(256, seed=0)
    assert torch.equal(x, x2)
    c, _ = synthetic_cifar10(16)
    assert c.shape == (16, 3, 32, 32)
    t = synthetic_tokens(4, seq_len=32, vocab_size=100)
    assert t.shape == (4, 33) and int(t.max()) < 100
    # class prototypes are shared between seeds: nearest-prototype works across splits
    xa, ya = synthetic_mnist(2000, seed=0)
    xb, yb = synthetic_mnist(500, seed=1)
    protos = torch.stack([xa[ya == k].float().mean(0) for k in range(10)]).flatten(1)
    pred = torch.cdist(xb.float().flatten(1), protos).argmin(1)
    assert float((pred == yb).float().mean()) > 0.9


def test_mnist_loader_reads_idx_files_and_falls_back(tmp_path):
    import struct

    raw = tmp_path / "MNIST" / "raw"
    raw.mkdir(parents=True)
    n = 50
    imgs = np.random.default_rng(0).integers(0, 256, size=(n, 28, 28), dtype=np.uint8)
    labs = np.random.default_rng(1).integers(0, 10, size=n, dtype=np.uint8)
    (raw / "train-images-idx3-ubyte").write_bytes(struct.pack(">HBBIII", 0, 8, 3, n, 28, 28) + imgs.tobytes())
    (raw / "train-labels-idx1-ubyte").write_bytes(struct.pack(">HBBI", 0, 8, 1, n) + labs.tobytes())
    assert np.array_equal(read_idx(raw / "train-images-idx3-ubyte"), imgs)
    loader = load_mnist_data(tmp_path, batch_size=10, train=True, download=False, subset_fraction=0.4)
    assert len(loader.dataset) == 20
    xb, yb = next(iter(loader))
    assert xb.shape == (10, 1, 28, 28) and xb.dtype == torch.float32 and yb.dtype == torch.int64
    with pytest.raises(RuntimeError):
        load_mnist_data(tmp_path / "nowhere", batch_size=10, download=False)
    fb = load_mnist_data(tmp_path / "nowhere", 10, download=False, synthetic_fallback=True,
                         subset_fraction=0.01)
    assert len(fb.dataset) == 600
