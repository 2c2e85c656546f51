"""MNIST classifiers.

``MNISTModel`` has the reference's architecture and state-dict keys
(nanofed/models/mnist.py:6-28): conv1 1->32 k3, ReLU, conv2 32->64 k3, ReLU,
2x2 max-pool, dropout .25, fc1 9216->128, ReLU, dropout .5, fc2 128->10,
log-softmax.  1,199,882 parameters in 8 tensors.  This ``nn.Module`` is the
portable definition (CPU, checkpoints, HTTP path, numerical oracle); on a
GPU the federated engine trains the same parameters with fused HIP kernels
(``nanofed_amd.engine.mnist_cnn``).

``MNISTMLP`` is the 784-H-10 two-layer perceptron used by the CPU/HTTP
plumbing configuration.
"""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class MNISTModel(nn.Module):
    def __init__(self, dropout: bool = True) -> None:
        """``dropout=False`` keeps the two dropout layers as identities (p = 0): a
        deterministic network for numerics comparisons; the default is the reference's."""
        super().__init__()
        self.conv1 = nn.Conv2d(1, 32, 3, 1)
        self.conv2 = nn.Conv2d(32, 64, 3, 1)
        self.dropout1 = nn.Dropout(0.25 if dropout else 0.0)
        self.dropout2 = nn.Dropout(0.5 if dropout else 0.0)
        self.fc1 = nn.Linear(9216, 128)
        self.fc2 = nn.Linear(128, 10)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        x = F.relu(self.conv1(x))
        x = F.relu(self.conv2(x))
        x = F.max_pool2d(x, 2)
        return torch.flatten(self.dropout1(x), 1)

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        x = F.relu(self.fc1(self.features(x)))
        return self.fc2(self.dropout2(x))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.log_softmax(self.logits(x), dim=1)


class MNISTMLP(nn.Module):
    """784 -> hidden -> 10 perceptron returning log-probabilities."""

    def __init__(self, hidden: int = 128) -> None:
        super().__init__()
        self.fc1 = nn.Linear(784, hidden)
        self.fc2 = nn.Linear(hidden, 10)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = F.relu(self.fc1(torch.flatten(x, 1)))
        return F.log_softmax(self.fc2(x), dim=1)
