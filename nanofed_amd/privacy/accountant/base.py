"""Accountant interface and the (epsilon, delta) spend record
(reference: nanofed/privacy/accountant/base.py:8-53)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Protocol

from ..config import PrivacyConfig


@dataclass(frozen=True)
class PrivacySpent:
    """Cumulative privacy loss."""

    epsilon_spent: float
    delta_spent: float

    def validate(self, config: PrivacyConfig) -> bool:
        """True while both components are within ``config``'s budget."""
        return (
            self.epsilon_spent <= config.epsilon
            and self.delta_spent <= config.delta
        )


class PrivacyAccountant(Protocol):
    def get_privacy_spent(self) -> PrivacySpent: ...
    def add_noise_event(self, sigma: float, samples: int) -> None: ...
    def validate_budget(self, config: PrivacyConfig) -> bool: ...


def check_event(sigma: float, samples: int) -> None:
    """Shared argument check; the messages are part of the API."""
    if samples <= 0:
        raise ValueError("Number of samples must be positive")
    if sigma <= 0:
        raise ValueError("Noise multiplier must be positive")


class BasePrivacyAccountant(ABC):
    """Common state: the budget, the last computed spend, an event counter."""

    def __init__(self, config: PrivacyConfig) -> None:
        self._config = config
        self._privacy_spent = PrivacySpent(0.0, 0.0)
        self._event_count = 0

    @abstractmethod
    def _compute_privacy_spent(self) -> PrivacySpent: ...

    def _compat_sampling_rate(self, samples: int) -> float:
        """Sampling rate as the reference defines it.

        ``q = min(samples / max_gradient_norm, 1)`` - the clipping bound
        doubles as the "dataset size".  Callers depend on the resulting
        closed forms, so the formula is kept as is
        (reference: accountant/gaussian.py:23-25, rdp.py:79-81).
        """
        return min(float(samples) / float(self._config.max_gradient_norm), 1.0)

    def get_privacy_spent(self) -> PrivacySpent:
        return self._compute_privacy_spent()

    def validate_budget(self, config: PrivacyConfig | None = None) -> bool:
        budget = config or self._config
        return bool(self.get_privacy_spent().validate(budget))
