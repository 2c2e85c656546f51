"""Differentially-private SGD trainer
(reference: nanofed/trainer/private.py:16-154).

Two gradient-sanitisation modes:

``clip_mode="batch"`` (default, the reference's behaviour)
    the batch-mean gradient is clipped to ``max_gradient_norm`` with
    ``clip_grad_norm_`` semantics and N(0, (sigma*C)^2) noise is added to it.

``clip_mode="per_sample"`` (DP-SGD as in Abadi et al. 2016)
    every example's gradient is clipped to ``C`` on its own, the clipped
    gradients are summed, N(0, (sigma*C)^2) noise is added to the sum and the
    result is divided by the batch size.  Per-example gradients come from
    ``torch.func`` (vmap over a functional call), so this works for any
    module without per-layer hooks.  The MNIST engine in
    ``nanofed_amd.engine`` implements the same step natively on the GPU
    without materialising the per-example gradients.

In both modes one accountant event ``(sigma, batch_size)`` is recorded per
batch.  Unlike the reference, callbacks passed to the constructor are kept,
batches are moved to the configured device, and ``train_epoch`` applies the
privacy step to every batch (the reference inherits the non-private epoch
loop).
"""

from __future__ import annotations

from typing import Literal, Sequence

import torch
import torch.nn as nn

from nanofed_amd.privacy.accountant import GaussianAccountant, PrivacySpent
from nanofed_amd.privacy.config import PrivacyConfig
from nanofed_amd.privacy.noise import GaussianNoiseGenerator

from .base import Callback, TrainingConfig, TrainingMetrics
from .torch import TorchTrainer

ClipMode = Literal["batch", "per_sample"]


class PrivateTrainer(TorchTrainer):
    def __init__(
        self,
        training_config: TrainingConfig,
        privacy_config: PrivacyConfig,
        accountant: GaussianAccountant | None = None,
        noise_generator: GaussianNoiseGenerator | None = None,
        callbacks: list[Callback] | None = None,
        clip_mode: ClipMode = "batch",
    ) -> None:
        super().__init__(training_config, callbacks)
        if clip_mode not in ("batch", "per_sample"):
            raise ValueError(f"Unknown clip_mode: {clip_mode}")
        self._privacy_config = privacy_config
        self._accountant = accountant or GaussianAccountant(privacy_config)
        self._noise_gen = noise_generator or GaussianNoiseGenerator()
        self._clip_mode: ClipMode = clip_mode

    # -- gradient sanitisation --------------------------------------------
    def _clip_gradients(self, model: nn.Module) -> None:
        """Scale all gradients so their global L2 norm is at most C."""
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        if not grads:
            return
        max_norm = self._privacy_config.max_gradient_norm
        total = torch.linalg.vector_norm(
            torch.stack(list(torch._foreach_norm(grads)))
        )
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        torch._foreach_mul_(grads, coef)

    def _add_noise(self, model: nn.Module, batch_size: int) -> None:
        """Add N(0, (sigma*C)^2) to every gradient and record the event."""
        cfg = self._privacy_config
        std = cfg.noise_multiplier * cfg.max_gradient_norm
        for p in model.parameters():
            if p.grad is None:
                continue
            noise = self._noise_gen.generate(tuple(p.grad.shape), scale=std)
            p.grad.add_(noise.to(p.grad.device, p.grad.dtype))
        self._accountant.add_noise_event(
            sigma=cfg.noise_multiplier, samples=batch_size
        )

    def _apply_privacy_to_gradients(
        self, model: nn.Module, batch_size: int
    ) -> None:
        self._clip_gradients(model)
        self._add_noise(model, batch_size)

    # -- per-sample path ---------------------------------------------------
    def _per_sample_backward(
        self, net: nn.Module, data: torch.Tensor, target: torch.Tensor
    ) -> tuple[torch.Tensor, torch.Tensor]:
        from torch.func import functional_call, grad, vmap

        names = [n for n, p in net.named_parameters() if p.requires_grad]
        params = {n: p.detach() for n, p in net.named_parameters()}
        buffers = {n: b.detach() for n, b in net.named_buffers()}

        def example_loss(p, x, y):
            out = functional_call(net, (p, buffers), (x.unsqueeze(0),))
            return self.compute_loss(out, y.unsqueeze(0)), out.squeeze(0)

        per_grads, outputs = vmap(
            grad(example_loss, has_aux=True), in_dims=(None, 0, 0),
            randomness="different",
        )(params, data, target)

        c = self._privacy_config.max_gradient_norm
        sq = torch.zeros(data.shape[0], device=data.device)
        for n in names:
            sq += per_grads[n].flatten(1).float().pow(2).sum(dim=1)
        coef = torch.clamp(c / (sq.sqrt() + 1e-6), max=1.0)
        for n, p in net.named_parameters():
            if n not in per_grads:
                continue
            g = per_grads[n]
            shaped = coef.view(-1, *([1] * (g.dim() - 1))).to(g.dtype)
            p.grad = (g * shaped).sum(dim=0)
        with torch.no_grad():
            loss = self.compute_loss(outputs, target)
        return outputs, loss

    def _forward_backward(self, net, data, target):
        if self._clip_mode == "per_sample":
            return self._per_sample_backward(net, data, target)
        return super()._forward_backward(net, data, target)

    def _after_backward(self, net: nn.Module, batch_size: int) -> None:
        if self._clip_mode == "per_sample":
            # the per-example clip already happened; noise goes on the SUM,
            # then the mean is taken
            self._add_noise(net, batch_size)
            grads = [p.grad for p in net.parameters() if p.grad is not None]
            torch._foreach_div_(grads, float(batch_size))
        else:
            self._apply_privacy_to_gradients(net, batch_size)

    # -- public API ---------------------------------------------------------
    def train_batch(
        self,
        model: nn.Module,
        batch: Sequence[torch.Tensor],
        optimizer: torch.optim.Optimizer,
    ) -> TrainingMetrics:
        """One DP-SGD step on ``batch = (inputs, targets)``."""
        inputs, targets = batch
        inputs = inputs.to(self._device)
        targets = targets.to(self._device)
        n = len(inputs)

        optimizer.zero_grad()
        outputs, loss = self._forward_backward(model, inputs, targets)
        self._after_backward(model, n)
        optimizer.step()

        return TrainingMetrics(
            loss=float(loss.detach()),
            accuracy=float(self.compute_accuracy(outputs.detach(), targets)),
            epoch=0,
            batch=0,
            samples_processed=n,
        )

    def get_privacy_spent(self) -> PrivacySpent:
        return self._accountant.get_privacy_spent()

    def validate_privacy_budget(self) -> bool:
        return self._accountant.validate_budget()
