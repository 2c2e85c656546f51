"""Linear-composition accountant for the Gaussian mechanism
(reference: nanofed/privacy/accountant/gaussian.py:7-48).

epsilon_total = sum_i c * q_i / sigma_i with c = sqrt(2 ln(1.25/delta)).
The running total is kept incrementally (O(1) per event); the additions
happen in event order, so the value equals a left-to-right sum over the
event list.
"""

from __future__ import annotations

import math

from ..config import PrivacyConfig
from .base import BasePrivacyAccountant, PrivacySpent, check_event


class GaussianAccountant(BasePrivacyAccountant):
    def __init__(self, config: PrivacyConfig) -> None:
        super().__init__(config)
        self._c = math.sqrt(2 * math.log(1.25 / config.delta))
        self._q = 0.0
        self._noise_multipliers: list[float] = []
        self._sample_rates: list[float] = []
        self._epsilon_total = 0.0

    def add_noise_event(self, sigma: float, samples: int) -> None:
        check_event(sigma, samples)
        self._q = self._compat_sampling_rate(samples)
        self._noise_multipliers.append(sigma)
        self._sample_rates.append(self._q)
        self._epsilon_total += self._c * self._q / sigma
        self._event_count += 1
        self._compute_privacy_spent()

    def _compute_privacy_spent(self) -> PrivacySpent:
        if self._event_count == 0:
            self._privacy_spent = PrivacySpent(0.0, 0.0)
        else:
            self._privacy_spent = PrivacySpent(
                epsilon_spent=self._epsilon_total,
                delta_spent=self._config.delta,
            )
        return self._privacy_spent
