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

With ``clip_mode="per_sample"``, an ``MNISTModel`` on a HIP device, plain SGD and
batches of up to 64 images, ``train_batch`` runs that step on the hand-written DP-SGD
kernels (``nanofed_amd.trainer.native.NativeDPStep``: per-example norms from one
backward pass, clip factors, a re-weighted pass, Philox Gaussian noise and the SGD
update fused into the last kernels) - chosen by the ``native`` argument exactly as in
``TorchTrainer`` and reported in ``TrainingMetrics.backend``.  Where the noise is drawn
in PyTorch and the model sits on a device, the default generator is the device one
(``DeviceGaussianNoiseGenerator``): no host ``randn`` and no host-to-device copy per
parameter and step.  Passing a generator keeps that generator.

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
from .torch import NativeChoice, TorchTrainer

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
        native: NativeChoice = "auto",
    ) -> None:
        super().__init__(training_config, callbacks, native=native)
        if clip_mode not in ("batch", "per_sample"):
            raise ValueError(f"Unknown clip_mode: {clip_mode}")
        self._privacy_config = privacy_config
        self._accountant = accountant or GaussianAccountant(privacy_config)
        # the noise is drawn where the gradients live: on a device, from the Philox stream
        # (a generator handed in by the caller is used as it is)
        self._device_noise = False
        if noise_generator is None and self._device.type == "cuda" and torch.cuda.is_available():
            from nanofed_amd.privacy.noise import DeviceGaussianNoiseGenerator

            noise_generator = DeviceGaussianNoiseGenerator(device=self._device)
            self._device_noise = True
        elif noise_generator is not None:
            self._device_noise = type(noise_generator).__name__ == "DeviceGaussianNoiseGenerator"
        self._noise_gen = noise_generator or GaussianNoiseGenerator()
        self._clip_mode: ClipMode = clip_mode
        self._native_dp = None  # NativeDPStep, built at the first native batch

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

        if self._native_batch_route(model, inputs, targets, optimizer):
            return self._native_train_batch(model, inputs, targets, optimizer)
        self.last_backend = "torch"

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

    # -- native route -------------------------------------------------------
    def train_epoch(self, model, dataloader, optimizer, epoch):
        """One epoch of DP-SGD steps.  When the call is the native step's (see
        ``train_batch``) and the loader is one the native epoch covers
        (``nanofed_amd.trainer.native``: random batches of 64 from a dataset that kept its
        uint8 images), the whole epoch is one hipGraph of native DP steps - the weights
        are copied in and out once per epoch instead of once per batch; one accountant
        event is recorded per batch either way."""
        import os

        from . import native
        from .base import _notify_epoch_start

        reason = None
        if self._native is False:
            reason = "native=False"
        elif self._clip_mode != "per_sample":
            reason = "clip_mode is 'batch'"
        elif not self._device_noise:
            reason = "a host noise generator was passed"
        else:
            reason = native.native_epoch_unsupported_reason(self, model, dataloader, optimizer)
            if reason is None and dataloader.batch_size != 64:
                reason = "the native DP-SGD step is built for batches of 64"
        if reason is None and self._native == "auto" and os.environ.get(
                "NANOFED_NATIVE_TRAINER", "1") == "0":
            reason = "switched off by NANOFED_NATIVE_TRAINER=0"
        if reason is not None:
            if self._native is True:
                from nanofed_amd.core.exceptions import DeviceError

                raise DeviceError(f"native=True, but {reason}")
            self.last_backend = "torch"
            return super().train_epoch(model, dataloader, optimizer, epoch)
        from functools import partial

        from nanofed_amd.engine.config import DPConfig

        cfg = self._privacy_config
        make = partial(
            native.NativeEpochSession, seed=int(getattr(self._noise_gen, "_seed", 0)),
            dp=DPConfig(max_gradient_norm=cfg.max_gradient_norm,
                        noise_multiplier=cfg.noise_multiplier, epsilon=cfg.epsilon,
                        delta=cfg.delta, mode="per_sample"))
        session = native.native_session_for(self, dataloader, optimizer, factory=make)
        _notify_epoch_start(self._callbacks, epoch)
        metrics = session.train_epoch(model, epoch)
        n, b = session.samples, int(dataloader.batch_size)
        for size in [b] * (n // b) + ([n % b] if n % b else []):
            self._accountant.add_noise_event(sigma=cfg.noise_multiplier, samples=size)
        self.last_backend = "native"
        return metrics

    def _native_batch_route(self, model, inputs, targets, optimizer) -> bool:
        import os

        from . import native

        if self._native is False:
            return False
        reason = native.native_dp_batch_unsupported_reason(self, model, inputs, targets, optimizer)
        if self._native is True and reason is not None:
            from nanofed_amd.core.exceptions import DeviceError

            raise DeviceError(f"native=True, but {reason}")
        if self._native == "auto" and os.environ.get("NANOFED_NATIVE_TRAINER", "1") == "0":
            return False
        return reason is None

    def _native_train_batch(self, model, inputs, targets, optimizer) -> TrainingMetrics:
        from . import native

        cfg = self._privacy_config
        if self._native_dp is None:
            self._logger.info(
                "train_batch runs on the native DP-SGD kernels (per-sample clipping, bf16 "
                "compute, Gaussian noise drawn on the device)"
            )
            seed = getattr(self._noise_gen, "_seed", 0)
            with native._NATIVE_LOCK:
                self._native_dp = native.NativeDPStep(
                    self._device, cfg.max_gradient_norm, cfg.noise_multiplier, int(seed))
        group = optimizer.param_groups[0]
        n = len(inputs)
        loss, acc = self._native_dp.step(
            model, inputs.to(torch.float32), targets, float(group["lr"]),
            float(group.get("weight_decay", 0.0)))
        self._accountant.add_noise_event(sigma=cfg.noise_multiplier, samples=n)
        self.last_backend = "native"
        return TrainingMetrics(loss=loss, accuracy=acc, epoch=0, batch=0, samples_processed=n,
                               backend="native")

    def get_privacy_spent(self) -> PrivacySpent:
        return self._accountant.get_privacy_spent()

    def validate_privacy_budget(self) -> bool:
        return self._accountant.validate_budget()
