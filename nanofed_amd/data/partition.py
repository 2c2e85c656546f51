"""Client sharding of a labelled dataset.

The reference hands every client an independent random subset, so shards
overlap (nanofed/data/mnist.py:30-36).  These partitioners produce disjoint
shards: ``iid_partition`` (uniform random split) and ``dirichlet_partition``
(label-skewed split; class proportions per client are drawn from
Dirichlet(alpha), small alpha = strongly non-IID).
Both are deterministic functions of ``seed``.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np


def iid_partition(
    num_samples: int, num_clients: int, seed: int = 0
) -> list[np.ndarray]:
    """Split ``range(num_samples)`` into ``num_clients`` near-equal random parts."""
    if num_clients <= 0:
        raise ValueError("num_clients must be positive")
    rng = np.random.default_rng(seed)
    perm = rng.permutation(num_samples)
    return [np.sort(part) for part in np.array_split(perm, num_clients)]


def sized_partition(
    num_samples: int, sizes: Sequence[int], seed: int = 0
) -> list[np.ndarray]:
    """Disjoint random shards of the given sizes (e.g. 12000/8000/4000)."""
    if sum(sizes) > num_samples:
        raise ValueError("Requested shard sizes exceed the dataset size")
    rng = np.random.default_rng(seed)
    perm = rng.permutation(num_samples)
    out, start = [], 0
    for size in sizes:
        out.append(np.sort(perm[start : start + size]))
        start += size
    return out


def dirichlet_partition(
    labels: np.ndarray,
    num_clients: int,
    alpha: float,
    seed: int = 0,
    min_size: int = 1,
    max_tries: int = 100,
) -> list[np.ndarray]:
    """Label-skewed disjoint split.

    For every class the samples of that class are divided among the clients
    in proportions drawn from Dirichlet(alpha * 1).  The draw is repeated (up
    to ``max_tries`` times) until every client holds at least ``min_size``
    samples.
    """
    if alpha <= 0:
        raise ValueError("alpha must be positive")
    labels = np.asarray(labels)
    classes = np.unique(labels)
    rng = np.random.default_rng(seed)
    for _ in range(max_tries):
        shards: list[list[np.ndarray]] = [[] for _ in range(num_clients)]
        for c in classes:
            idx = np.flatnonzero(labels == c)
            rng.shuffle(idx)
            props = rng.dirichlet(np.full(num_clients, alpha))
            cuts = (np.cumsum(props)[:-1] * len(idx)).astype(int)
            for k, part in enumerate(np.split(idx, cuts)):
                shards[k].append(part)
        result = [
            np.sort(np.concatenate(s)) if s else np.empty(0, dtype=np.int64)
            for s in shards
        ]
        if min(len(r) for r in result) >= min_size:
            return result
    raise RuntimeError(
        f"Could not draw a Dirichlet({alpha}) partition with >= {min_size} "
        f"samples per client in {max_tries} tries"
    )


def label_histogram(
    labels: np.ndarray, shards: Sequence[np.ndarray], num_classes: int
) -> np.ndarray:
    """``[num_clients, num_classes]`` table of label counts per shard."""
    labels = np.asarray(labels)
    return np.stack(
        [np.bincount(labels[s], minlength=num_classes) for s in shards]
    )
