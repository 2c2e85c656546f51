"""Federated averaging (reference: nanofed/server/aggregator/fedavg.py:10-125).

theta <- sum_k w_k * theta_k with w_k = n_k / sum n.  Every state-dict entry
(buffers included) is averaged in fp32, as in the reference.

Two execution paths produce the same numbers:

* arena path - when every update's ``model_state`` is a view dictionary over a
  flat parameter arena (``ArenaStateDict``) of one layout, the K client arenas
  are reduced by a single fused weighted-sum kernel over the flat buffers
  (``nanofed_amd.ops.weighted_sum``), regardless of how many tensors the model
  has;
* generic path - per-key accumulation for plain dictionaries of tensors or
  nested lists (the HTTP wire format).
"""

from __future__ import annotations

from typing import Any, Sequence

import torch

from nanofed_amd.core import ModelProtocol, ModelUpdate
from nanofed_amd.utils import get_current_time, log_exec

from .base import AggregationResult, BaseAggregator, normalise, sample_counts


def as_float_tensor(value: Any) -> torch.Tensor:
    if isinstance(value, torch.Tensor):
        return value.detach()
    return torch.tensor(value, dtype=torch.float32)


def weighted_state_sum(
    states: Sequence[dict[str, Any]], weights: Sequence[float]
) -> dict[str, torch.Tensor]:
    """sum_k w_k * state_k, key by key, accumulated in fp32."""
    flats = [getattr(s, "flat", None) for s in states]
    if all(isinstance(f, torch.Tensor) for f in flats) and len(
        {(f.shape, f.device) for f in flats}
    ) == 1 and len({id(getattr(s, "layout", None)) for s in states}) == 1:
        from nanofed_amd import ops

        out = ops.weighted_sum(list(flats), list(weights))
        return states[0].with_flat(out)  # type: ignore[attr-defined]

    result: dict[str, torch.Tensor] = {}
    for key in states[0]:
        acc = as_float_tensor(states[0][key]).to(torch.float32) * weights[0]
        for state, w in zip(states[1:], weights[1:]):
            acc.add_(as_float_tensor(state[key]).to(acc.device), alpha=w)
        result[key] = acc
    return result


class FedAvgAggregator(BaseAggregator[ModelProtocol]):
    def _to_tensor(self, data: Any) -> torch.Tensor:
        if isinstance(data, torch.Tensor):
            return data.clone().detach()
        return torch.tensor(data, dtype=torch.float32)

    @log_exec
    def aggregate(
        self, model: ModelProtocol, updates: Sequence[ModelUpdate]
    ) -> AggregationResult[ModelProtocol]:
        self._validate_updates(updates)
        weights = self._compute_weights(updates)

        merged = weighted_state_sum(
            [u["model_state"] for u in updates], weights
        )
        model.load_state_dict(merged)

        metrics = self._aggregate_metrics(updates, weights)
        self._current_round += 1
        return AggregationResult(
            model=model,
            round_number=self._current_round,
            num_clients=len(updates),
            timestamp=get_current_time(),
            metrics=metrics,
        )

    def _aggregate_metrics(
        self, updates: Sequence[ModelUpdate], weights: list[float]
    ) -> dict[str, float]:
        """Weighted mean of every numeric metric over the clients reporting it."""
        num: dict[str, float] = {}
        den: dict[str, float] = {}
        for update, w in zip(updates, weights):
            for key, value in update["metrics"].items():
                if isinstance(value, bool) or not isinstance(value, (int, float)):
                    continue
                num[key] = num.get(key, 0.0) + float(value) * w
                den[key] = den.get(key, 0.0) + w
        return {k: num[k] / den[k] for k in num if den[k] != 0}

    def _compute_weights(self, updates: Sequence[ModelUpdate]) -> list[float]:
        counts = sample_counts(updates, self._logger)
        weights = normalise(counts)
        self._logger.debug(f"Client sample counts: {counts}")
        self._logger.debug(f"Computed weights: {weights}")
        return weights
