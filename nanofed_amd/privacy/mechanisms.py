"""Clip-and-noise mechanisms for model updates
(reference: nanofed/privacy/mechanisms.py:17-174).

``add_noise(parameters, batch_size)``:
  1. scales the whole update so that its global L2 norm is at most
     ``max_gradient_norm`` (coefficient ``min(C / (norm + 1e-6), 1)``),
  2. adds noise of scale ``noise_multiplier * C / batch_size`` to every tensor
     (one ``generate(shape, scale=...)`` call per tensor),
  3. records one accountant event ``(sigma=noise_multiplier, samples=batch_size)``.

When the update is a view dictionary over one contiguous parameter arena
(anything exposing a 1-D ``.flat`` tensor, see
``nanofed_amd.runtime.arena.ArenaStateDict``) and the noise generator is a
device Philox generator, steps 1-2 run as two fused kernels over the flat
buffer (sum-of-squares reduce, then scale + Philox noise) instead of a Python
loop over tensors.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from enum import Enum, auto
from typing import Any, Protocol, TypeAlias, TypedDict

import torch

from nanofed_amd.utils.logger import Logger

from .accountant import GaussianAccountant, PrivacySpent
from .config import PrivacyConfig
from .noise import GaussianNoiseGenerator

ModelState: TypeAlias = dict[str, torch.Tensor]

_NORM_EPS = 1e-6


class PrivacyType(Enum):
    CENTRAL = auto()
    LOCAL = auto()


class PrivacyMetrics(TypedDict):
    epsilon_spent: float
    delta_spent: float
    noise_scale: float
    clip_ratio: float


class PrivacyMechanism(Protocol):
    def add_noise(
        self, parameters: ModelState, batch_size: int
    ) -> ModelState: ...

    def get_privacy_spent(self) -> PrivacySpent: ...

    @property
    def privacy_type(self) -> PrivacyType: ...


@dataclass(slots=True, frozen=True)
class UpdateMetadata:
    total_norm: float
    clipped_norm: float
    num_parameters: int
    noise_scale: float


def global_l2_norm(parameters: ModelState) -> torch.Tensor:
    """L2 norm of the concatenation of all tensors, without concatenating."""
    flat = getattr(parameters, "flat", None)
    if isinstance(flat, torch.Tensor):
        return torch.linalg.vector_norm(flat.float())
    tensors = [p.detach().float() for p in parameters.values()]
    if not tensors:
        return torch.zeros(())
    per_tensor = torch._foreach_norm(tensors)
    return torch.linalg.vector_norm(torch.stack(list(per_tensor)))


class BasePrivacyMechanism(ABC):
    def __init__(
        self,
        config: PrivacyConfig,
        accountant: GaussianAccountant | None = None,
        noise_generator: GaussianNoiseGenerator | None = None,
    ) -> None:
        self._config = config
        self._accountant = accountant or GaussianAccountant(config)
        self._noise_gen = noise_generator or GaussianNoiseGenerator()
        self._logger = Logger()

    @property
    @abstractmethod
    def privacy_type(self) -> PrivacyType: ...

    def _compute_noise_scale(self, batch_size: int) -> float:
        cfg = self._config
        return cfg.noise_multiplier * cfg.max_gradient_norm / batch_size

    def _clip_update(
        self, parameters: ModelState, max_norm: float
    ) -> tuple[ModelState, UpdateMetadata]:
        norm = global_l2_norm(parameters)
        coef = torch.clamp(max_norm / (norm + _NORM_EPS), max=1.0)
        clipped = {name: p * coef.to(p.device) for name, p in parameters.items()}
        total = float(norm)
        meta = UpdateMetadata(
            total_norm=total,
            clipped_norm=total * float(coef),
            num_parameters=sum(p.numel() for p in parameters.values()),
            noise_scale=self._config.noise_multiplier,
        )
        return clipped, meta

    def _fused_arena_path(
        self, parameters: ModelState, noise_scale: float
    ) -> tuple[ModelState, UpdateMetadata] | None:
        """Clip + noise over a flat arena in two kernels, when applicable."""
        flat = getattr(parameters, "flat", None)
        fused = getattr(self._noise_gen, "clip_and_noise_flat", None)
        if not isinstance(flat, torch.Tensor) or fused is None:
            return None
        out_flat, total = fused(
            flat, self._config.max_gradient_norm, noise_scale
        )
        coef = min(self._config.max_gradient_norm / (total + _NORM_EPS), 1.0)
        meta = UpdateMetadata(
            total_norm=total,
            clipped_norm=total * coef,
            num_parameters=int(flat.numel()),
            noise_scale=self._config.noise_multiplier,
        )
        return parameters.with_flat(out_flat), meta  # type: ignore[attr-defined]

    def add_noise(self, parameters: ModelState, batch_size: int) -> ModelState:
        noise_scale = self._compute_noise_scale(batch_size)

        fused = self._fused_arena_path(parameters, noise_scale)
        if fused is not None:
            noised, meta = fused
        else:
            clipped, meta = self._clip_update(
                parameters, self._config.max_gradient_norm
            )
            noised = {}
            for name, tensor in clipped.items():
                noise = self._noise_gen.generate(
                    tuple(tensor.shape), scale=noise_scale
                )
                noised[name] = tensor + noise.to(tensor.device)

        self._accountant.add_noise_event(
            sigma=self._config.noise_multiplier, samples=batch_size
        )
        self._logger.debug(
            "Applied privacy mechanism: "
            f"norm={meta.total_norm:.3f}->{meta.clipped_norm:.3f}, "
            f"noise={noise_scale:.3f}"
        )
        return noised

    def get_privacy_spent(self) -> PrivacySpent:
        return self._accountant.get_privacy_spent()

    def validate_budget(self) -> bool:
        return self._accountant.validate_budget()


class CentralPrivacyMechanism(BasePrivacyMechanism):
    """Noise calibrated to the number of contributions in the aggregate."""

    @property
    def privacy_type(self) -> PrivacyType:
        return PrivacyType.CENTRAL


class LocalPrivacyMechanism(BasePrivacyMechanism):
    """Each update is privatised on its own, i.e. with ``batch_size = 1``."""

    @property
    def privacy_type(self) -> PrivacyType:
        return PrivacyType.LOCAL

    def add_noise(self, parameters: ModelState, batch_size: int) -> ModelState:
        return super().add_noise(parameters, batch_size=1)


class PrivacyMechanismFactory:
    _registry = {
        PrivacyType.CENTRAL: CentralPrivacyMechanism,
        PrivacyType.LOCAL: LocalPrivacyMechanism,
    }

    @staticmethod
    def create(
        privacy_type: PrivacyType, config: PrivacyConfig, **kwargs: Any
    ) -> BasePrivacyMechanism:
        cls = PrivacyMechanismFactory._registry.get(privacy_type)
        if cls is None:
            raise ValueError(f"Unknown privacy type: {privacy_type}")
        return cls(config, **kwargs)
