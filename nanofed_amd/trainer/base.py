"""Local-training loop and its configuration
(reference: nanofed/trainer/base.py:15-198).

Differences from the reference loop, all deliberate:

* the value returned by ``train_epoch`` is the epoch aggregate (mean loss and
  accuracy over the batches that actually ran, cumulative
  ``samples_processed``) - the reference hands that aggregate to the
  callbacks but returns the last batch's numbers;
* the batch count used for the means is the number of batches that ran, also
  when ``max_batches`` cuts the epoch short;
* loss and correct-prediction counts are accumulated in device tensors and
  read back only when a callback or a progress line needs a host value, so an
  epoch without callbacks synchronises with the GPU once per ``log_interval``
  batches and once at the end instead of twice per batch;
* ``validate`` runs a held-out evaluation pass (the reference declares it in
  ``TrainerProtocol`` and never implements it).
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Generic, Iterable, Protocol, Sized, TypeVar, cast

import torch
import torch.nn as nn

from nanofed_amd.core import ModelProtocol
from nanofed_amd.utils import Logger, log_exec

M = TypeVar("M", bound=ModelProtocol)


@dataclass(slots=True, frozen=True)
class TrainingConfig:
    epochs: int
    batch_size: int
    learning_rate: float
    device: str = "cpu"
    max_batches: int | None = None
    log_interval: int = 10


@dataclass(slots=True)
class TrainingMetrics:
    loss: float
    accuracy: float
    epoch: int
    batch: int
    samples_processed: int
    # which implementation produced the numbers: "torch" (PyTorch ops in the dtype of the
    # model) or "native" (the hand-written bf16 kernels of nanofed_amd.engine); not part of
    # the wire format (``to_dict``)
    backend: str = "torch"

    def to_dict(self) -> dict[str, float | int]:
        """The three fields that travel to the server with an update."""
        return {
            "loss": self.loss,
            "accuracy": self.accuracy,
            "samples_processed": self.samples_processed,
        }


class Callback(Protocol):
    """Training hooks.  ``on_eopch_start`` is the reference's public
    (misspelled) hook name; trainers call ``on_epoch_start`` as well when a
    callback defines it."""

    def on_eopch_start(self, epoch: int) -> None: ...
    def on_epoch_end(self, epoch: int, metrics: TrainingMetrics) -> None: ...
    def on_batch_end(self, batch: int, metrics: TrainingMetrics) -> None: ...


def _notify_epoch_start(callbacks: Iterable[Callback], epoch: int) -> None:
    for cb in callbacks:
        hook = getattr(cb, "on_eopch_start", None) or getattr(
            cb, "on_epoch_start", None
        )
        if hook is not None:
            hook(epoch)


class BaseTrainer(ABC, Generic[M]):
    def __init__(
        self,
        config: TrainingConfig,
        callbacks: list[Callback] | None = None,
    ) -> None:
        self._config = config
        self._callbacks = callbacks or []
        self._logger = Logger()
        self._device = torch.device(config.device)

    @abstractmethod
    def compute_loss(
        self, output: torch.Tensor, target: torch.Tensor
    ) -> torch.Tensor: ...

    @abstractmethod
    def compute_accuracy(
        self, output: torch.Tensor, target: torch.Tensor
    ) -> float: ...

    def count_correct(
        self, output: torch.Tensor, target: torch.Tensor
    ) -> torch.Tensor | None:
        """Number of correct predictions as a device scalar, or ``None`` when
        a subclass only provides the host-float ``compute_accuracy``."""
        return None

    def _forward_backward(
        self, net: nn.Module, data: torch.Tensor, target: torch.Tensor
    ) -> tuple[torch.Tensor, torch.Tensor]:
        """Forward pass, loss, and gradient accumulation into ``.grad``."""
        output = net(data)
        loss = self.compute_loss(output, target)
        loss.backward()
        return output, loss

    def _after_backward(self, net: nn.Module, batch_size: int) -> None:
        """Hook between backward and the optimizer step (DP trainers clip and
        perturb the gradients here)."""

    @log_exec
    def train_epoch(
        self,
        model: M,
        dataloader: Iterable,
        optimizer: torch.optim.Optimizer,
        epoch: int,
    ) -> TrainingMetrics:
        net = cast(nn.Module, model)
        net.train()
        cfg = self._config
        dev = self._device

        loss_sum = torch.zeros((), device=dev, dtype=torch.float32)
        acc_sum = torch.zeros((), device=dev, dtype=torch.float32)
        samples = 0
        batches_run = 0
        last_idx = -1
        need_host_every_batch = bool(self._callbacks)

        _notify_epoch_start(self._callbacks, epoch)

        for batch_idx, (data, target) in enumerate(dataloader):
            if cfg.max_batches is not None and batch_idx >= cfg.max_batches:
                break
            data = data.to(dev, non_blocking=True)
            target = target.to(dev, non_blocking=True)

            optimizer.zero_grad()
            output, loss = self._forward_backward(net, data, target)
            n = int(len(data))
            self._after_backward(net, n)
            optimizer.step()

            correct = self.count_correct(output.detach(), target)
            if correct is None:
                batch_acc_t = torch.tensor(
                    float(self.compute_accuracy(output.detach(), target)),
                    device=dev,
                )
            else:
                batch_acc_t = correct.to(torch.float32) / n
            loss_d = loss.detach().to(torch.float32)
            loss_sum += loss_d
            acc_sum += batch_acc_t
            samples += n
            batches_run += 1
            last_idx = batch_idx

            log_now = batch_idx % cfg.log_interval == 0
            if need_host_every_batch or log_now:
                loss_f, acc_f = float(loss_d), float(batch_acc_t)
                if need_host_every_batch:
                    snap = TrainingMetrics(
                        loss=loss_f,
                        accuracy=acc_f,
                        epoch=epoch,
                        batch=batch_idx,
                        samples_processed=samples,
                    )
                    for cb in self._callbacks:
                        cb.on_batch_end(batch_idx, snap)
                if log_now:
                    total = self._dataset_size(dataloader)
                    pct = 100.0 * samples / total if total else 0.0
                    self._logger.info(
                        f"Train Epoch: {epoch} "
                        f"[{samples}/{total} ({pct:.0f}%)] "
                        f"Loss: {loss_f:.6f} "
                        f"Accuracy: {acc_f:.4f}"
                    )

        denom = float(max(batches_run, 1))
        result = TrainingMetrics(
            loss=float(loss_sum) / denom,
            accuracy=float(acc_sum) / denom,
            epoch=epoch,
            batch=max(last_idx, 0),
            samples_processed=samples,
        )
        for cb in self._callbacks:
            cb.on_epoch_end(epoch, result)
        return result

    @torch.no_grad()
    def validate(self, model: M, dataloader: Iterable) -> dict[str, float]:
        """Held-out evaluation: mean loss and accuracy over ``dataloader``."""
        net = cast(nn.Module, model)
        was_training = net.training
        net.eval()
        dev = self._device
        loss_sum = torch.zeros((), device=dev, dtype=torch.float64)
        correct_sum = torch.zeros((), device=dev, dtype=torch.float64)
        seen = 0
        for data, target in dataloader:
            data, target = data.to(dev), target.to(dev)
            out = net(data)
            n = int(len(data))
            loss_sum += self.compute_loss(out, target).double() * n
            correct = self.count_correct(out, target)
            if correct is None:
                correct_sum += self.compute_accuracy(out, target) * n
            else:
                correct_sum += correct.double()
            seen += n
        net.train(was_training)
        if seen == 0:
            return {"loss": 0.0, "accuracy": 0.0, "samples": 0.0}
        return {
            "loss": float(loss_sum) / seen,
            "accuracy": float(correct_sum) / seen,
            "samples": float(seen),
        }

    @staticmethod
    def _dataset_size(dataloader: Iterable) -> int:
        dataset = getattr(dataloader, "dataset", None)
        if dataset is not None:
            try:
                return len(cast(Sized, dataset))
            except TypeError:
                return 0
        return 0
