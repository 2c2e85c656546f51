"""Device-resident client shards.

The reference feeds each client from a ``DataLoader`` and copies every batch
host -> device (nanofed/trainer/base.py:141; loader: nanofed/data/mnist.py:9-40).
MNIST is 47 MB as uint8 and CIFAR-10 150 MB, so the whole dataset is placed
in HBM once, as uint8, and a batch is an index gather on the device; the
``(x/255 - mean)/std`` normalisation happens at batch-assembly time (inside
the first convolution kernel on the native path).

Per-epoch shuffling is a device ``randperm`` per client, seeded by
``(seed, client id, round, epoch)`` so the sample order of a client does not
depend on how many GPUs the clients are spread over.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from nanofed_amd.data import (
    dirichlet_partition,
    iid_partition,
    sized_partition,
    synthetic_cifar10,
    synthetic_cifar10_hard,
    synthetic_mnist,
    synthetic_mnist_hard,
    synthetic_tokens,
)
from nanofed_amd.data.cifar import CIFAR10_MEAN, CIFAR10_STD, load_cifar10_arrays
from nanofed_amd.data.tokens import load_token_sequences
from nanofed_amd.data.mnist import MNIST_MEAN, MNIST_STD, load_mnist_arrays

from .config import EngineConfig


@dataclass
class FederatedData:
    """Training pool, per-client index shards and a held-out split."""

    kind: str  # "image" | "tokens"
    inputs: torch.Tensor  # uint8 [N,C,H,W] or int64 [N,S+1]
    labels: torch.Tensor | None
    shards: list[np.ndarray]  # one index array per (global) client
    eval_inputs: torch.Tensor
    eval_labels: torch.Tensor | None
    mean: tuple[float, ...] = (0.0,)
    std: tuple[float, ...] = (1.0,)


IMAGE_DATASETS = ("synthetic_mnist", "synthetic_mnist_hard", "mnist", "synthetic_cifar10",
                  "synthetic_cifar10_hard", "cifar10")


def _sizes(cfg: EngineConfig) -> list[int]:
    spc = cfg.samples_per_client
    if isinstance(spc, int):
        return [spc] * cfg.num_clients
    sizes = [int(s) for s in spc]
    if len(sizes) != cfg.num_clients:
        raise ValueError("samples_per_client must list one size per client")
    return sizes


def build_federated_data(cfg: EngineConfig) -> FederatedData:
    """Generate / load the dataset named by ``cfg`` and shard it over
    ``cfg.num_clients`` clients.  Deterministic in ``cfg.seed``; every rank
    builds the same object and keeps only its clients' shards on its GPU."""
    sizes = _sizes(cfg)
    total = sum(sizes)
    if cfg.dataset in IMAGE_DATASETS:
        if cfg.dataset in ("synthetic_mnist", "synthetic_mnist_hard"):
            make = synthetic_mnist if cfg.dataset == "synthetic_mnist" else synthetic_mnist_hard
            x, y = make(total, seed=cfg.seed)
            ex, ey = make(cfg.eval_samples, seed=cfg.seed + 1)
            mean, std = (MNIST_MEAN,), (MNIST_STD,)
        elif cfg.dataset == "mnist":
            x, y = load_mnist_arrays(cfg.data_dir, train=True, synthetic_fallback=True)
            ex, ey = load_mnist_arrays(cfg.data_dir, train=False, synthetic_fallback=True)
            ex, ey = ex[: cfg.eval_samples], ey[: cfg.eval_samples]
            mean, std = (MNIST_MEAN,), (MNIST_STD,)
        elif cfg.dataset == "cifar10":
            # the binary batches under cfg.data_dir (never the pickled version)
            x, y = load_cifar10_arrays(cfg.data_dir, train=True, synthetic_fallback=True)
            ex, ey = load_cifar10_arrays(cfg.data_dir, train=False, synthetic_fallback=True)
            ex, ey = ex[: cfg.eval_samples], ey[: cfg.eval_samples]
            mean, std = CIFAR10_MEAN, CIFAR10_STD
        else:
            make = synthetic_cifar10 if cfg.dataset == "synthetic_cifar10" else synthetic_cifar10_hard
            x, y = make(total, seed=cfg.seed)
            ex, ey = make(cfg.eval_samples, seed=cfg.seed + 1)
            mean, std = CIFAR10_MEAN, CIFAR10_STD
        if total > len(x):
            raise ValueError(
                f"clients ask for {total} samples, dataset has {len(x)}"
            )
        if cfg.partition == "iid":
            shards = sized_partition(len(x), sizes, seed=cfg.seed)
        elif cfg.partition == "dirichlet":
            pool = sized_partition(len(x), [total], seed=cfg.seed)[0]
            parts = dirichlet_partition(
                y.numpy()[pool], cfg.num_clients, cfg.dirichlet_alpha,
                seed=cfg.seed, min_size=cfg.batch_size,
            )
            shards = [pool[p] for p in parts]
        else:
            raise ValueError(f"unknown partition {cfg.partition!r}")
        return FederatedData("image", x, y, shards, ex, ey, mean, std)
    if cfg.dataset in ("synthetic_tokens", "tokens"):
        vocab = int(cfg.model_kwargs.get("vocab_size", 50257))
        if cfg.dataset == "tokens":
            # <data_dir>/train.bin and val.bin (flat token ids, memory-mapped); a missing
            # file falls back to the synthetic stream with a warning, like the image sets
            toks = load_token_sequences(cfg.data_dir, "train", cfg.seq_len, total, vocab,
                                        seed=cfg.seed, dtype=cfg.token_dtype,
                                        synthetic_fallback=True)
            ev = load_token_sequences(cfg.data_dir, "val", cfg.seq_len, max(cfg.eval_samples, 1),
                                      vocab, seed=cfg.seed + 1, dtype=cfg.token_dtype,
                                      synthetic_fallback=True)
        else:
            toks = synthetic_tokens(total, cfg.seq_len, vocab, seed=cfg.seed,
                                    active_vocab=cfg.token_active_vocab)
            ev = synthetic_tokens(
                max(cfg.eval_samples, 1), cfg.seq_len, vocab, seed=cfg.seed + 1,
                table_seed=cfg.seed, active_vocab=cfg.token_active_vocab,
            )
        shards = (
            iid_partition(total, cfg.num_clients, seed=cfg.seed)
            if len(set(sizes)) == 1
            else sized_partition(total, sizes, seed=cfg.seed)
        )
        return FederatedData("tokens", toks, None, shards, ev, None)
    raise ValueError(f"unknown dataset {cfg.dataset!r}")


class DeviceShards:
    """The shards of this rank's clients, resident on its device."""

    def __init__(
        self,
        data: FederatedData,
        client_ids: list[int],
        device: torch.device,
        batch_size: int,
        seed: int,
        drop_last: bool = False,
    ) -> None:
        self.kind = data.kind
        self.device = device
        self.batch_size = batch_size
        self.seed = seed
        self.drop_last = drop_last
        self.client_ids = client_ids
        # keep only the rows this rank trains on (re-indexed), in HBM
        local = [np.asarray(data.shards[c]) for c in client_ids]
        rows = np.concatenate(local) if local else np.zeros(0, dtype=np.int64)
        sel = torch.from_numpy(rows.astype(np.int64))
        self.inputs = data.inputs[sel].contiguous().to(device)
        self.labels = (
            data.labels[sel].contiguous().to(device) if data.labels is not None else None
        )
        self.offsets = np.concatenate([[0], np.cumsum([len(s) for s in local])]).astype(int)
        self.sizes = [len(s) for s in local]
        shape = (1, -1, 1, 1)
        self.mean = torch.tensor(data.mean, dtype=torch.float32, device=device).view(shape)
        self.std = torch.tensor(data.std, dtype=torch.float32, device=device).view(shape)
        # The shuffles are drawn on the host, whatever the device: the same permutations on
        # every device, and no chain of tiny sort kernels competing with training for a
        # hardware queue (the host is idle while an epoch's launches run).
        self._gen = torch.Generator()

    def steps_per_epoch(self, k: int) -> int:
        n, b = self.sizes[k], self.batch_size
        return n // b if self.drop_last else (n + b - 1) // b

    def max_steps(self) -> int:
        return max((self.steps_per_epoch(k) for k in range(len(self.sizes))), default=0)

    def host_permutation(self, k: int, round_idx: int, epoch: int, pinned: bool = False) -> torch.Tensor:
        """Row indices (into ``self.inputs``) of client k's samples, shuffled: an int64
        tensor in host memory (page-locked on request), a function of (seed, client id,
        round, epoch) only."""
        cid = self.client_ids[k]
        mix = (self.seed * 1000003 + cid * 7919 + round_idx * 104729 + epoch) & 0x7FFFFFFFFFFF
        self._gen.manual_seed(mix)
        out = torch.empty(self.sizes[k], dtype=torch.int64, pin_memory=pinned)
        torch.randperm(self.sizes[k], generator=self._gen, out=out)
        return out.add_(int(self.offsets[k]))

    def epoch_permutation(self, k: int, round_idx: int, epoch: int) -> torch.Tensor:
        """The same on the training device.  The copy from page-locked memory does not block
        the host (the caching host allocator keeps the buffer until the copy has run)."""
        if self.device.type != "cuda":
            return self.host_permutation(k, round_idx, epoch)
        host = self.host_permutation(k, round_idx, epoch, pinned=True)
        return host.to(self.device, non_blocking=True)

    def batch(self, rows: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """Assemble a normalised batch (torch path)."""
        if self.kind == "image":
            x = self.inputs[rows].to(torch.float32).div_(255.0)
            x = x.sub_(self.mean).div_(self.std)
            return x, self.labels[rows]
        toks = self.inputs[rows]
        return toks[:, :-1], toks[:, 1:]
