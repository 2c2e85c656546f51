"""Renyi-DP accountants.

``RDPAccountant`` reproduces the reference's bookkeeping
(nanofed/privacy/accountant/rdp.py:11-115): per event and order alpha it
adds q^2 * alpha / (2 sigma^2), and converts with
eps = min_alpha [ RDP(alpha) + ln(1/delta) / (alpha - 1) ].

``SubsampledRDPAccountant`` is an addition: the integer-order bound for the
Poisson-subsampled Gaussian mechanism (Mironov, Talwar, Zhang 2019, Sec. 3.3)
with a true sampling rate q = batch / dataset_size, for users who want a
meaningful DP-SGD guarantee rather than the compat closed form.
"""

from __future__ import annotations

import math
from typing import Sequence

import numpy as np

from ..config import PrivacyConfig
from ..exceptions import PrivacyError
from .base import BasePrivacyAccountant, PrivacySpent, check_event

_DEFAULT_ORDERS = (1.5, 2.0, 2.5, 3.0, 4.0, 8.0, 16.0, 32.0, 64.0)


class RDPAccountant(BasePrivacyAccountant):
    def __init__(
        self, config: PrivacyConfig, orders: Sequence[float] | None = None
    ) -> None:
        super().__init__(config)
        self._orders = np.asarray(
            list(orders) if orders else list(_DEFAULT_ORDERS), dtype=np.float64
        )
        if self._orders.size == 0:
            raise PrivacyError("Must specify at least one RDP order")
        if not bool(np.all(self._orders > 1.0)):
            raise PrivacyError("All RDP orders must be > 1.0")
        # one accumulator per order, vectorised
        self._rdp = np.zeros_like(self._orders)

    @property
    def _rdp_budget(self) -> dict[float, float]:
        return {float(a): float(r) for a, r in zip(self._orders, self._rdp)}

    def _compute_rdp_gaussian(
        self, sigma: float, sampling_rate: float
    ) -> dict[float, float]:
        inc = (sampling_rate**2) * self._orders / (2 * sigma**2)
        return {float(a): float(v) for a, v in zip(self._orders, inc)}

    def add_noise_event(self, sigma: float, samples: int) -> None:
        check_event(sigma, samples)
        q = self._compat_sampling_rate(samples)
        self._rdp += (q**2) * self._orders / (2 * sigma**2)
        self._event_count += 1
        self._compute_privacy_spent()

    def _compute_privacy_spent(self) -> PrivacySpent:
        delta = self._config.delta
        eps = self._rdp + math.log(1 / delta) / (self._orders - 1)
        self._privacy_spent = PrivacySpent(
            epsilon_spent=float(eps.min()), delta_spent=delta
        )
        return self._privacy_spent


def _log_add(a: float, b: float) -> float:
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    hi, lo = (a, b) if a > b else (b, a)
    return hi + math.log1p(math.exp(lo - hi))


def _log_comb(n: int, k: int) -> float:
    return math.lgamma(n + 1) - math.lgamma(k + 1) - math.lgamma(n - k + 1)


def subsampled_gaussian_rdp(q: float, sigma: float, alpha: int) -> float:
    """RDP of order ``alpha`` (integer >= 2) of the sampled Gaussian mechanism."""
    if q <= 0:
        return 0.0
    if q >= 1.0:
        return alpha / (2 * sigma**2)
    log_a = -math.inf
    for i in range(alpha + 1):
        term = (
            _log_comb(alpha, i)
            + i * math.log(q)
            + (alpha - i) * math.log1p(-q)
            + (i * i - i) / (2 * sigma**2)
        )
        log_a = _log_add(log_a, term)
    return log_a / (alpha - 1)


class SubsampledRDPAccountant(BasePrivacyAccountant):
    """DP-SGD accountant with Poisson-subsampling amplification."""

    def __init__(
        self,
        config: PrivacyConfig,
        dataset_size: int,
        orders: Sequence[int] | None = None,
    ) -> None:
        super().__init__(config)
        if dataset_size <= 0:
            raise PrivacyError("dataset_size must be positive")
        self._dataset_size = int(dataset_size)
        self._orders = [int(a) for a in (orders or range(2, 65))]
        if not self._orders or min(self._orders) < 2:
            raise PrivacyError("Integer RDP orders must be >= 2")
        self._rdp = [0.0] * len(self._orders)
        self._cache: dict[tuple[float, float], list[float]] = {}

    def add_noise_event(self, sigma: float, samples: int) -> None:
        check_event(sigma, samples)
        q = min(samples / self._dataset_size, 1.0)
        inc = self._cache.get((q, sigma))
        if inc is None:
            inc = [subsampled_gaussian_rdp(q, sigma, a) for a in self._orders]
            self._cache[(q, sigma)] = inc
        self._rdp = [r + d for r, d in zip(self._rdp, inc)]
        self._event_count += 1

    def epsilon_after(self, events: int, sigma: float, samples: int) -> float:
        """Epsilon spent by ``events`` identical noise events, in closed form: RDP
        composes additively, so ``events`` steps cost ``events`` times one step at
        every order.  Nothing is recorded; a trainer whose step count is a function
        of the round number asks this once per round instead of replaying events."""
        check_event(sigma, samples)
        if events <= 0:
            return 0.0
        q = min(samples / self._dataset_size, 1.0)
        inc = self._cache.get((q, sigma))
        if inc is None:
            inc = [subsampled_gaussian_rdp(q, sigma, a) for a in self._orders]
            self._cache[(q, sigma)] = inc
        delta = self._config.delta
        return min(
            events * d + math.log(1 / delta) / (a - 1) for d, a in zip(inc, self._orders)
        )

    def _compute_privacy_spent(self) -> PrivacySpent:
        if self._event_count == 0:
            return PrivacySpent(0.0, 0.0)
        delta = self._config.delta
        eps = min(
            r + math.log(1 / delta) / (a - 1)
            for r, a in zip(self._rdp, self._orders)
        )
        self._privacy_spent = PrivacySpent(eps, delta)
        return self._privacy_spent
