"""Privacy-aware aggregation
(reference: nanofed/server/aggregator/privacy.py:20-346).

CENTRAL: the server clips and noises every client state
(``mechanism.add_noise(state, batch_size=K)``) before combining.
LOCAL: clients privatise their own updates and must report ``privacy_spent``;
their sample-proportional weights are re-scaled by the epsilon they spent
(w_k ~ n_k * eps_k).  With THRESHOLD secure aggregation the (noised) states are
summed; otherwise they are weight-averaged.

Validation failures raise ``ValueError`` with the reference's messages.
``aggregate`` advances ``current_round`` (the reference leaves it at 0 for
ever) and the PrivacySpent -> dict conversion maps delta from ``delta_spent``.
"""

from __future__ import annotations

from enum import Enum, auto
from typing import Any, Protocol, Sequence, cast

import torch
from pydantic import ConfigDict, Field

from nanofed_amd.core import ModelProtocol, ModelUpdate
from nanofed_amd.privacy.accountant import PrivacySpent
from nanofed_amd.privacy.config import PrivacyConfig
from nanofed_amd.privacy.mechanisms import (
    BasePrivacyMechanism,
    PrivacyMechanismFactory,
    PrivacyType,
)
from nanofed_amd.utils import Logger

from .base import AggregationResult, BaseAggregator, normalise, sample_counts
from .fedavg import weighted_state_sum


class SecureAggregationType(Enum):
    NONE = auto()
    THRESHOLD = auto()
    HOMOMORPHIC = auto()


class PrivacyAwareAggregationConfig(PrivacyConfig):
    model_config = ConfigDict(frozen=True, arbitrary_types_allowed=True)

    privacy_type: PrivacyType = Field(default=PrivacyType.CENTRAL)
    secure_aggregation: SecureAggregationType = Field(
        default=SecureAggregationType.NONE
    )
    min_clients: int = Field(default=1, ge=1)
    dropout_tolerance: float = Field(default=0.0, ge=0.0, le=1.0)
    clip_norm: float = Field(default=1.0, gt=0.0)


class SecureAggregationProtocol(Protocol):
    def aggregate_shares(
        self, shares: Sequence[dict[str, torch.Tensor]]
    ) -> dict[str, torch.Tensor]: ...

    def verify_shares(
        self, shares: Sequence[dict[str, torch.Tensor]]
    ) -> bool: ...


class ThresholdSecureAggregation:
    """Sum of shares, refused below ``min_clients`` contributors."""

    def __init__(self, min_clients: int) -> None:
        self._min_clients = min_clients
        self._logger = Logger()

    def aggregate_shares(
        self, shares: Sequence[dict[str, torch.Tensor]]
    ) -> dict[str, torch.Tensor]:
        if len(shares) < self._min_clients:
            raise ValueError(
                f"Not enough clients: {len(shares)} < {self._min_clients}"
            )
        # an unweighted sum is the weighted sum with every weight = 1
        return dict(weighted_state_sum(list(shares), [1.0] * len(shares)))

    def verify_shares(self, shares: Sequence[dict[str, torch.Tensor]]) -> bool:
        if len(shares) < self._min_clients:
            return False
        reference = {k: tuple(v.shape) for k, v in shares[0].items()}
        for share in shares[1:]:
            if share.keys() != reference.keys():
                return False
            if any(tuple(share[k].shape) != s for k, s in reference.items()):
                return False
        return True


def _epsilon_of(spent: Any) -> float:
    if isinstance(spent, PrivacySpent):
        as_dict = {"epsilon": spent.epsilon_spent, "delta": spent.delta_spent}
    elif isinstance(spent, dict):
        as_dict = spent
    else:
        raise TypeError(
            "privacy_spent should be a dict or PrivacySpent "
            f"instance, got {type(spent)}"
        )
    return float(as_dict.get("epsilon", 1.0))


class PrivacyAwareAggregator(BaseAggregator):
    def __init__(
        self,
        config: PrivacyAwareAggregationConfig,
        privacy_mechanism: BasePrivacyMechanism | None = None,
        secure_aggregation: SecureAggregationProtocol | None = None,
    ) -> None:
        super().__init__()
        self._config = config
        self._privacy_mech = privacy_mechanism or PrivacyMechanismFactory.create(
            config.privacy_type, config=config
        )
        self._secure_agg = secure_aggregation
        if (
            secure_aggregation is None
            and config.secure_aggregation == SecureAggregationType.THRESHOLD
        ):
            self._secure_agg = ThresholdSecureAggregation(config.min_clients)

    def _validate_updates(self, updates: Sequence[ModelUpdate]) -> None:
        if not updates:
            raise ValueError("No updates provided")
        if len(updates) < self._config.min_clients:
            raise ValueError(
                f"Not enough clients: {len(updates)} < {self._config.min_clients}"
            )
        rounds = {u.get("round_number") for u in updates}
        if len(rounds) != 1:
            raise ValueError("Updates from different rounds")
        keys = updates[0]["model_state"].keys()
        if any(u["model_state"].keys() != keys for u in updates):
            raise ValueError("Inconsistent model architectures")
        if self._config.privacy_type == PrivacyType.LOCAL:
            for u in updates:
                if u.get("privacy_spent") is None:
                    raise ValueError(
                        f"Missing privacy budget for client {u['client_id']}"
                    )

    def _process_local_updates(
        self, updates: Sequence[ModelUpdate]
    ) -> Sequence[ModelUpdate]:
        return list(updates)

    def _process_central_updates(
        self, updates: Sequence[ModelUpdate]
    ) -> Sequence[ModelUpdate]:
        k = len(updates)
        out = []
        for u in updates:
            noised = self._privacy_mech.add_noise(u["model_state"], batch_size=k)
            out.append(cast(ModelUpdate, {**u, "model_state": noised}))
        return out

    def _compute_weights(self, updates: Sequence[ModelUpdate]) -> list[float]:
        weights = normalise(sample_counts(updates, self._logger))
        if self._config.privacy_type == PrivacyType.LOCAL:
            eps = [
                _epsilon_of(u.get("privacy_spent", {"epsilon": 1.0, "delta": 1e-5}))
                for u in updates
            ]
            if sum(eps) > 0:
                weights = normalise(
                    [w * e for w, e in zip(weights, normalise(eps))]
                )
        self._logger.debug(f"Computed weights: {weights}")
        return weights

    def _aggregate_metrics(
        self,
        updates: Sequence[ModelUpdate],
        weights: list[float] | None = None,
    ) -> dict[str, float]:
        """Weighted sum of every numeric metric (absent = 0) plus the
        mechanism's cumulative (epsilon, delta)."""
        if not updates:
            return {}
        if weights is None:
            weights = normalise(
                [u["metrics"].get("samples_processed", 1) for u in updates]
            )
        keys: set[str] = set()
        for u in updates:
            for name, value in (u.get("metrics") or {}).items():
                if isinstance(value, (int, float)):
                    keys.add(name)
        out = {
            name: sum(
                u["metrics"].get(name, 0) * w for u, w in zip(updates, weights)
            )
            for name in keys
        }
        if self._privacy_mech is not None:
            spent = self._privacy_mech.get_privacy_spent()
            out["privacy_epsilon"] = spent.epsilon_spent
            out["privacy_delta"] = spent.delta_spent
        return out

    def aggregate(
        self, model: ModelProtocol, updates: Sequence[ModelUpdate]
    ) -> AggregationResult:
        self._validate_updates(updates)
        if self._config.privacy_type == PrivacyType.LOCAL:
            processed = self._process_local_updates(updates)
        else:
            processed = self._process_central_updates(updates)
        states = [u["model_state"] for u in processed]

        if self._secure_agg is not None:
            if not self._secure_agg.verify_shares(states):
                raise ValueError("Invalid shares for secure aggregation")
            merged = self._secure_agg.aggregate_shares(states)
        else:
            merged = weighted_state_sum(states, self._compute_weights(processed))

        model.load_state_dict(merged)
        self._current_round += 1
        return AggregationResult(
            model=model,
            round_number=self._current_round,
            num_clients=len(updates),
            timestamp=self._get_timestamp(),
            metrics=self._aggregate_metrics(processed),
        )
