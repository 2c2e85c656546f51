"""Cross-entropy classification trainer (reference: nanofed/trainer/torch.py:7-22).

``train_epoch`` keeps the reference's signature and semantics
(nanofed/trainer/base.py:115-198); when the call is exactly the step the
hand-written MNIST kernels implement (``nanofed_amd.trainer.native``), the epoch
runs on them instead of PyTorch ops.
"""

from __future__ import annotations

import os
from typing import Any, Union

import torch
import torch.nn.functional as F

from .base import BaseTrainer, Callback, TrainingConfig, TrainingMetrics, _notify_epoch_start


NativeChoice = Union[bool, str]


class TorchTrainer(BaseTrainer):
    """``native`` chooses the implementation of ``train_epoch`` on a HIP device:

    ``"auto"`` (default)
        the hand-written bf16 kernels when the call is exactly the step they implement
        (``nanofed_amd.trainer.native.native_epoch_supported``), PyTorch ops otherwise.
        The native step computes in bf16 with fp32 master weights, draws its own shuffle
        and dropout streams, and is what makes the call fast; the choice is reported in
        ``TrainingMetrics.backend``, in ``trainer.last_backend`` and in one log line per
        data loader.  ``NANOFED_NATIVE_TRAINER=0`` turns "auto" into ``False``.
    ``True``
        the native kernels, or a ``DeviceError`` that says why the call is not theirs.
    ``False``
        always PyTorch ops (the numerics of the model's own dtype).
    """

    def __init__(self, config: TrainingConfig, callbacks: list[Callback] | None = None,
                 native: NativeChoice = "auto") -> None:
        super().__init__(config, callbacks)
        if native not in (True, False, "auto"):
            raise ValueError(f"native must be True, False or 'auto', not {native!r}")
        self._native: NativeChoice = native
        self.last_backend = "torch"

    def _native_route(self, model: Any, dataloader: Any, optimizer: Any) -> bool:
        from . import native

        if self._native is False:
            return False
        reason = native.native_epoch_unsupported_reason(self, model, dataloader, optimizer)
        if self._native is True and reason is not None:
            from nanofed_amd.core.exceptions import DeviceError

            raise DeviceError(f"native=True, but {reason}")
        if self._native == "auto" and os.environ.get("NANOFED_NATIVE_TRAINER", "1") == "0":
            return False
        return reason is None

    def train_epoch(self, model: Any, dataloader: Any, optimizer: Any, epoch: int) -> TrainingMetrics:
        from . import native

        # exact type: subclasses (DP trainers) hook into the per-batch loop
        if type(self) is TorchTrainer and self._native_route(model, dataloader, optimizer):
            session = native.native_session_for(self, dataloader, optimizer)
            _notify_epoch_start(self._callbacks, epoch)
            self.last_backend = "native"
            return session.train_epoch(model, epoch)
        self.last_backend = "torch"
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
