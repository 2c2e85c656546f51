"""Aggregation strategy base (reference: nanofed/server/aggregator/base.py:14-82)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from datetime import datetime
from typing import Generic, Sequence, TypeVar

from nanofed_amd.core.exceptions import AggregationError
from nanofed_amd.core.interfaces import ModelProtocol
from nanofed_amd.core.types import ModelUpdate
from nanofed_amd.utils import Logger, get_current_time

T = TypeVar("T", bound=ModelProtocol)


@dataclass(slots=True, frozen=True)
class AggregationResult(Generic[T]):
    model: T
    round_number: int
    num_clients: int
    timestamp: datetime
    metrics: dict[str, float]


def sample_counts(updates: Sequence[ModelUpdate], logger: Logger) -> list[float]:
    """Per-client sample count: ``num_samples``, else ``samples_processed``,
    else 1.0 with a warning (reference: fedavg.py:101-117)."""
    counts: list[float] = []
    for update in updates:
        metrics = update.get("metrics", {}) or {}
        n = metrics.get("num_samples") or metrics.get("samples_processed")
        if n is None:
            logger.warning(
                f"Client {update.get('client_id')} did not report sample "
                f"count. Using 1.0"
            )
            n = 1.0
        counts.append(float(n))
    return counts


def normalise(values: Sequence[float]) -> list[float]:
    total = sum(values)
    return [v / total for v in values]


class BaseAggregator(ABC, Generic[T]):
    """Subclass contract: ``aggregate(model, updates)`` and
    ``_compute_weights(updates)``."""

    def __init__(self) -> None:
        self._logger = Logger()
        self._current_round: int = 0
        self._weights_cache: dict[int, list[float]] = {}

    @property
    def current_round(self) -> int:
        return self._current_round

    def _get_timestamp(self) -> datetime:
        return get_current_time()

    def _validate_updates(self, updates: Sequence[ModelUpdate]) -> None:
        if not updates:
            raise AggregationError("No updates provided for aggregation")
        rounds = {u["round_number"] for u in updates}
        if len(rounds) != 1:
            raise AggregationError(f"Updates from different rounds: {rounds}")
        reference_keys = updates[0]["model_state"].keys()
        if any(u["model_state"].keys() != reference_keys for u in updates[1:]):
            raise AggregationError(
                "Inconsistent model architectures in updates."
            )

    @abstractmethod
    def aggregate(
        self, model: T, updates: Sequence[ModelUpdate]
    ) -> AggregationResult[T]: ...

    @abstractmethod
    def _compute_weights(self, updates: Sequence[ModelUpdate]) -> list[float]: ...
