"""Cross-entropy classification trainer (reference: nanofed/trainer/torch.py:7-22)."""

from __future__ import annotations

import torch
import torch.nn.functional as F

from .base import BaseTrainer


class TorchTrainer(BaseTrainer):
    def compute_loss(
        self, output: torch.Tensor, target: torch.Tensor
    ) -> torch.Tensor:
        # ``cross_entropy`` applies log-softmax itself; models that already
        # return log-probabilities are unaffected (log-softmax is idempotent).
        return F.cross_entropy(output, target)

    def count_correct(
        self, output: torch.Tensor, target: torch.Tensor
    ) -> torch.Tensor:
        return (output.argmax(dim=1) == target.view(-1)).sum()

    def compute_accuracy(
        self, output: torch.Tensor, target: torch.Tensor
    ) -> float:
        return int(self.count_correct(output, target)) / len(target)
