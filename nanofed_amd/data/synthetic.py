"""Synthetic datasets with the shapes of MNIST, CIFAR-10 and token streams.

There is no network in the build/bench environment, so benchmarks and tests
train on generated data of the right shape.  The image generators are
class-conditional (a fixed random low-frequency prototype per class, plus
per-sample noise and a random shift), so a model can actually learn them and
"final accuracy" is a meaningful sanity signal; held-out data is drawn from
the same distribution with a different seed.
"""

from __future__ import annotations

import numpy as np
import torch


def _prototypes(
    num_classes: int, channels: int, size: int, rng: np.random.Generator
) -> np.ndarray:
    low = rng.normal(size=(num_classes, channels, size // 4, size // 4))
    up = np.kron(low, np.ones((1, 1, 4, 4)))
    up = up[:, :, :size, :size]
    up -= up.min(axis=(1, 2, 3), keepdims=True)
    up /= up.max(axis=(1, 2, 3), keepdims=True) + 1e-9
    return up.astype(np.float32)


def synthetic_images(
    num_samples: int,
    num_classes: int = 10,
    channels: int = 1,
    size: int = 28,
    noise: float = 0.35,
    seed: int = 0,
    proto_seed: int = 1234,
    prototypes_per_class: int = 1,
    label_noise: float = 0.0,
    max_shift: int = 2,
) -> tuple[torch.Tensor, torch.Tensor]:
    """``(uint8 images [N,C,H,W], int64 labels [N])``.

    ``proto_seed`` fixes the class prototypes (shared by train and test
    splits); ``seed`` fixes the labels, noise and shifts.  With several
    ``prototypes_per_class`` a class is a mixture of patterns; ``label_noise`` is
    the fraction of samples whose label is replaced by a random class (a ceiling
    on the reachable accuracy, in training and held-out data alike).
    """
    ppc = max(1, prototypes_per_class)
    protos = _prototypes(
        num_classes * ppc, channels, size, np.random.default_rng(proto_seed)
    )
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, num_classes, size=num_samples)
    if ppc > 1:
        imgs = protos[labels * ppc + rng.integers(0, ppc, size=num_samples)]
    else:
        imgs = protos[labels]
    ms = max_shift
    shifts = rng.integers(-ms, ms + 1, size=(num_samples, 2))
    out = np.empty_like(imgs)
    for dy in range(-ms, ms + 1):
        for dx in range(-ms, ms + 1):
            sel = np.flatnonzero((shifts[:, 0] == dy) & (shifts[:, 1] == dx))
            if sel.size:
                out[sel] = np.roll(imgs[sel], (dy, dx), axis=(2, 3))
    out += rng.normal(scale=noise, size=out.shape).astype(np.float32)
    u8 = np.clip(out * 255.0, 0, 255).astype(np.uint8)
    if label_noise > 0:
        flip = rng.random(num_samples) < label_noise
        labels = np.where(flip, rng.integers(0, num_classes, size=num_samples), labels)
    return torch.from_numpy(u8), torch.from_numpy(labels.astype(np.int64))


def synthetic_mnist(num_samples: int = 60000, seed: int = 0):
    return synthetic_images(num_samples, 10, 1, 28, seed=seed)


def synthetic_mnist_hard(num_samples: int = 60000, seed: int = 0):
    """MNIST-shaped set that does not saturate in a round: three patterns per class,
    stronger pixel noise and 5 % random labels.  The CNN needs several hundred SGD steps
    to get past 0.6 held-out accuracy and cannot exceed about 0.95, so a regression in
    training shows up in the accuracy a run reports."""
    return synthetic_images(num_samples, 10, 1, 28, noise=0.45, seed=seed, prototypes_per_class=3,
                            label_noise=0.05)


def synthetic_cifar10(num_samples: int = 50000, seed: int = 0):
    return synthetic_images(num_samples, 10, 3, 32, seed=seed)


def synthetic_cifar10_hard(num_samples: int = 50000, seed: int = 0):
    """CIFAR-shaped set on which a ResNet cannot reach accuracy 1.0: four patterns per
    class, pixel noise of half the value range, shifts of up to 3 pixels and 8 % random
    labels (held-out accuracy is capped near 0.93), so a numerical regression of the
    training step moves the accuracy a run reports."""
    return synthetic_images(num_samples, 10, 3, 32, noise=0.5, seed=seed, prototypes_per_class=4,
                            label_noise=0.08, max_shift=3)


def synthetic_tokens(
    num_sequences: int, seq_len: int = 1024, vocab_size: int = 50257, seed: int = 0,
    table_seed: int | None = None, active_vocab: int | None = None,
) -> torch.Tensor:
    """``int64 [N, seq_len+1]`` token ids from a sparse first-order Markov
    chain (each token has 4 likely successors), so next-token loss can drop.

    ``table_seed`` fixes the successor table separately from the sequences: a
    held-out split drawn with another ``seed`` but the same ``table_seed`` follows
    the same chain, so evaluating on it measures what training learned.
    ``active_vocab`` confines the chain to the first ids of the vocabulary (the
    model keeps its full embedding / LM-head width): with few training tokens a
    chain over all 50257 ids cannot be learned, one over a few thousand can."""
    rng = np.random.default_rng(seed)
    table_rng = np.random.default_rng(seed if table_seed is None else table_seed)
    active = vocab_size if not active_vocab else min(int(active_vocab), vocab_size)
    succ = table_rng.integers(0, active, size=(active, 4))
    toks = np.empty((num_sequences, seq_len + 1), dtype=np.int64)
    toks[:, 0] = rng.integers(0, active, size=num_sequences)
    choice = rng.integers(0, 4, size=(num_sequences, seq_len))
    jump = rng.random(size=(num_sequences, seq_len)) < 0.1
    rand = rng.integers(0, active, size=(num_sequences, seq_len))
    for t in range(seq_len):
        nxt = succ[toks[:, t], choice[:, t]]
        toks[:, t + 1] = np.where(jump[:, t], rand[:, t], nxt)
    return torch.from_numpy(toks)
