"""Round orchestration (reference: nanofed/orchestration/coordinator.py:26-405).

A round: wait until ``required`` clients have submitted updates for the current
round (or time out), build ``ModelUpdate`` records, optionally screen them with
a validator, aggregate into the model manager's model, save a new model
version, write ``metrics/metrics_round_<id>.json`` and advance the round.

Behaviour that differs from the reference on purpose:

* ``privacy_spent`` is optional in a submitted update;
* the server's round counter is advanced together with the coordinator's, so
  clients fetch and submit the right round and stale updates are refused;
* the quorum is ``max(1, floor(min_clients * min_completion_rate))`` - never 0;
* clients that submit are tracked in ``ClientInfo`` records, so
  ``training_progress["active_clients"]`` is meaningful;
* an optional ``DefaultModelValidator`` drops invalid/anomalous updates before
  aggregation and an optional ``FaultTolerantCoordinator`` checkpoints every
  completed round (``checkpoints/round_<id>``).

The ``server`` only needs the transport surface used here (``_updates``,
``set_coordinator``, ``stop_training`` and optionally ``advance_round``), so the
HTTP server and the in-node transport are interchangeable.
"""

from __future__ import annotations

import asyncio
import json
import math
from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from typing import TYPE_CHECKING, Any, AsyncGenerator, Callable

import torch

from nanofed_amd.core import ModelManagerProtocol, ModelUpdate
from nanofed_amd.server import BaseAggregator
from nanofed_amd.utils import Logger, get_current_time, log_exec, monotonic_s

from .types import ClientInfo, RoundMetrics, RoundStatus, TrainingProgress

if TYPE_CHECKING:
    from nanofed_amd.communication import HTTPServer
    from nanofed_amd.server.fault_tolerance import FaultTolerantCoordinator
    from nanofed_amd.server.validation import DefaultModelValidator


@dataclass(slots=True, frozen=True)
class CoordinatorConfig:
    num_rounds: int
    min_clients: int
    min_completion_rate: float
    round_timeout: int
    base_dir: Path
    poll_interval: float = 0.05


def _to_tensor(value: Any) -> torch.Tensor:
    return value if isinstance(value, torch.Tensor) else torch.tensor(value)


def _parse_time(value: Any) -> datetime:
    if isinstance(value, datetime):
        return value
    try:
        return datetime.fromisoformat(value)
    except Exception:
        return get_current_time()


class Coordinator:
    def __init__(
        self,
        model_manager: ModelManagerProtocol,
        aggregator: BaseAggregator,
        server: "HTTPServer",
        config: CoordinatorConfig,
        validator: "DefaultModelValidator | None" = None,
        fault_tolerance: "FaultTolerantCoordinator | None" = None,
    ) -> None:
        self._model_manager = model_manager
        self._aggregator = aggregator
        self._server = server
        self._config = config
        self._validator = validator
        self._fault_tolerance = fault_tolerance
        self._logger = Logger()

        self._current_round: int = 0
        self._clients: dict[str, ClientInfo] = {}
        self._round_metrics: list[RoundMetrics] = []
        self._status = RoundStatus.INITIALIZED
        self._round_lock = asyncio.Lock()

        base = Path(config.base_dir)
        self._metrics_dir = base / "metrics"
        self._data_dir = base / "data"
        self._models_dir = base / "models"
        self._model_configs_dir = self._models_dir / "configs"
        self._model_weights_dir = self._models_dir / "models"
        self._setup_directories()

        self._model_manager.set_dirs(
            self._model_weights_dir, self._model_configs_dir
        )
        self._server.set_coordinator(self)
        self._sync_server_round()

    # -- accessors -----------------------------------------------------------
    @property
    def server(self) -> "HTTPServer":
        return self._server

    @property
    def data_dir(self) -> Path:
        return self._data_dir

    @property
    def model_manager(self) -> ModelManagerProtocol:
        return self._model_manager

    @property
    def current_round(self) -> int:
        return self._current_round

    @property
    def training_progress(self) -> TrainingProgress:
        return {
            "current_round": self._current_round,
            "total_rounds": self._config.num_rounds,
            "active_clients": len(self._clients),
            "global_metrics": self._get_global_metrics(),
            "status": self._status.name,
        }

    # -- helpers -------------------------------------------------------------
    def _setup_directories(self) -> None:
        with self._logger.context("coordinator.setup"):
            for directory in (
                self._metrics_dir,
                self._data_dir,
                self._model_configs_dir,
                self._model_weights_dir,
            ):
                try:
                    directory.mkdir(parents=True, exist_ok=True)
                    self._logger.info(f"Created directory: {directory}")
                except Exception as exc:
                    self._logger.error(
                        f"Failed to create directory {directory}: {exc}"
                    )
                    raise

    def _sync_server_round(self) -> None:
        advance = getattr(self._server, "advance_round", None)
        if callable(advance):
            advance(self._current_round)
        else:
            self._server._current_round = self._current_round

    def _get_global_metrics(self) -> dict[str, float]:
        """Unweighted mean of each aggregated metric over completed rounds."""
        series: dict[str, list[float]] = {}
        for rm in self._round_metrics:
            for key, value in rm.agg_metrics.items():
                series.setdefault(key, []).append(value)
        return {k: sum(v) / len(v) for k, v in series.items()}

    def _required_clients(self) -> int:
        cfg = self._config
        return max(1, math.floor(cfg.min_clients * cfg.min_completion_rate + 1e-9))

    async def _wait_for_clients(self, timeout: float) -> bool:
        with self._logger.context("coordinator"):
            need = self._required_clients()
            t0 = monotonic_s()
            last_seen = -1
            while monotonic_s() - t0 < timeout:
                have = len(self._server._updates)
                if have != last_seen:
                    last_seen = have
                    self._logger.info(
                        f"Client training progress: {have}/"
                        f"{self._config.min_clients} (need {need})"
                    )
                if have >= need:
                    self._logger.info(
                        f"Sufficient clients completed training: {have}/"
                        f"{self._config.min_clients}"
                    )
                    return True
                await asyncio.sleep(self._config.poll_interval)
            self._logger.error(
                "Timeout waiting for clients. "
                f"Got {len(self._server._updates)}/{self._config.min_clients} "
                f"(needed {need})"
            )
            return False

    def _save_metrics(
        self, metrics: RoundMetrics, client_metrics: list[dict]
    ) -> None:
        with self._logger.context(
            "coordinator.metrics", f"round_{metrics.round_id}"
        ):
            path = self._metrics_dir / f"metrics_round_{metrics.round_id}.json"
            record = {
                "round_id": metrics.round_id,
                "start_time": metrics.start_time.isoformat()
                if metrics.start_time else None,
                "end_time": metrics.end_time.isoformat()
                if metrics.end_time else None,
                "num_clients": metrics.num_clients,
                "agg_metrics": metrics.agg_metrics,
                "status": metrics.status.name,
                "client_metrics": client_metrics,
            }
            try:
                with path.open("w") as f:
                    json.dump(record, f, indent=4)
                self._logger.info(
                    f"Saved metrics for round {metrics.round_id} to {path}"
                )
            except Exception as exc:
                self._logger.error(
                    f"Failed to save metrics for round {metrics.round_id}: {exc}"
                )

    def _collect_updates(self) -> list[ModelUpdate]:
        collected: list[ModelUpdate] = []
        for raw in list(self._server._updates.values()):
            update = ModelUpdate(
                client_id=raw["client_id"],
                round_number=raw["round_number"],
                model_state={
                    k: _to_tensor(v) for k, v in raw["model_state"].items()
                } if not hasattr(raw["model_state"], "flat") else raw["model_state"],
                metrics=raw.get("metrics", {}),
                timestamp=_parse_time(raw.get("timestamp")),
                privacy_spent=raw.get("privacy_spent"),
            )
            collected.append(update)
            self._clients[update["client_id"]] = ClientInfo(
                client_id=update["client_id"],
                status="submitted",
                last_update=update["timestamp"],
                metrics=dict(update["metrics"]),
            )
        return collected

    def _screen(self, updates: list[ModelUpdate]) -> list[ModelUpdate]:
        """Drop updates the validator rejects (never all of them silently)."""
        if self._validator is None:
            return updates
        from nanofed_amd.server.validation import ValidationResult

        shapes = {
            k: v.shape for k, v in self._model_manager.model.state_dict().items()
        }
        kept = []
        for u in updates:
            others = [o for o in updates if o is not u]
            verdict = self._validator.validate_all(u, shapes, others)
            if verdict is ValidationResult.VALID:
                kept.append(u)
            else:
                self._logger.warning(
                    f"Rejected update from {u['client_id']}: {verdict.name}"
                )
        if not kept:
            raise RuntimeError("All client updates were rejected by validation")
        return kept

    # -- the round -----------------------------------------------------------
    @log_exec
    async def train_round(self) -> RoundMetrics:
        with self._logger.context("coordinator", f"round_{self._current_round}"):
            async with self._round_lock:
                try:
                    self._status = RoundStatus.IN_PROGRESS
                    start_time = get_current_time()
                    self._sync_server_round()

                    if not await self._wait_for_clients(
                        self._config.round_timeout
                    ):
                        self._status = RoundStatus.FAILED
                        raise TimeoutError(
                            f"Round {self._current_round} timed out waiting "
                            "for clients"
                        )

                    self._status = RoundStatus.AGGREGATING
                    updates = self._screen(self._collect_updates())
                    weights = self._aggregator._compute_weights(updates)
                    client_weights = {
                        u["client_id"]: w for u, w in zip(updates, weights)
                    }
                    client_metrics = [
                        {
                            "client_id": u["client_id"],
                            "metrics": u.get("metrics", {}),
                            "weight": client_weights[u["client_id"]],
                        }
                        for u in updates
                    ]

                    result = self._aggregator.aggregate(
                        self._model_manager.model, updates
                    )

                    version = self._model_manager.save_model(
                        config={
                            "round_id": self._current_round,
                            "client_metrics": client_metrics,
                            "client_weights": client_weights,
                            "start_time": start_time.isoformat(),
                            "status": self._status.name,
                            "num_clients": len(updates),
                        },
                        metrics=result.metrics,
                    )

                    finished = self._current_round
                    self._current_round += 1
                    self._status = RoundStatus.COMPLETED
                    metrics = RoundMetrics(
                        round_id=finished,
                        start_time=start_time,
                        end_time=get_current_time(),
                        num_clients=len(updates),
                        agg_metrics=result.metrics,
                        status=self._status,
                    )
                    self._round_metrics.append(metrics)
                    self._save_metrics(metrics, client_metrics)

                    if self._fault_tolerance is not None:
                        from nanofed_amd.server.fault_tolerance import RoundState

                        self._fault_tolerance.checkpoint_round(
                            round_id=finished,
                            client_updates={},
                            model_version=getattr(version, "version_id", ""),
                            state={
                                "model_state": dict(
                                    self._model_manager.model.state_dict()
                                ),
                                "round_id": finished,
                            },
                            round_state=RoundState.COMPLETED,
                        )

                    self._server._updates.clear()
                    self._sync_server_round()
                    return metrics
                except Exception as exc:
                    self._status = RoundStatus.FAILED
                    self._logger.error(
                        f"Error in round {self._current_round}: {exc}"
                    )
                    raise

    async def start_training(
        self,
        progress_callback: Callable[[TrainingProgress], None] | None = None,
    ) -> AsyncGenerator[RoundMetrics, None]:
        with self._logger.context("coordinator"):
            try:
                for _ in range(self._config.num_rounds):
                    metrics = await self.train_round()
                    if progress_callback:
                        progress_callback(self.training_progress)
                    yield metrics
                await self._server.stop_training()
            except Exception as exc:
                self._logger.error(f"Training failed: {exc}")
                raise
            finally:
                self._logger.info("Training completed")

    def recover(self, failure: Exception) -> bool:
        """Roll the global model back to the latest COMPLETED checkpoint after
        a recoverable failure.  Returns whether a rollback happened."""
        if self._fault_tolerance is None:
            return False
        restored = self._fault_tolerance.handle_failure(
            failure, self._current_round
        )
        if restored is None:
            return False
        meta, state = restored
        self._model_manager.model.load_state_dict(state["model_state"])
        self._current_round = meta.round_id + 1
        self._server._updates.clear()
        self._sync_server_round()
        self._status = RoundStatus.INITIALIZED
        return True
