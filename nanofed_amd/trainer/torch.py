"""Cross-entropy classification trainer (reference: nanofed/trainer/torch.py:7-22).

``train_epoch`` keeps the reference's signature and semantics
(nanofed/trainer/base.py:115-198); when the call is exactly the step the
hand-written MNIST kernels implement (``nanofed_amd.trainer.native``), the epoch
runs on them instead of PyTorch ops.
"""

from __future__ import annotations

from typing import Any

import torch
import torch.nn.functional as F

from .base import BaseTrainer, TrainingMetrics, _notify_epoch_start


class TorchTrainer(BaseTrainer):
    def train_epoch(self, model: Any, dataloader: Any, optimizer: Any, epoch: int) -> TrainingMetrics:
        from . import native

        # exact type: subclasses (DP trainers) hook into the per-batch loop
        if type(self) is TorchTrainer and native.native_epoch_supported(
            self, model, dataloader, optimizer
        ):
            session = native.native_session_for(self, dataloader, optimizer)
            _notify_epoch_start(self._callbacks, epoch)
            return session.train_epoch(model, epoch)
        return super().train_epoch(model, dataloader, optimizer, epoch)

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
