"""The coordinator's public surface on top of the device plane.

``Coordinator`` (reference: nanofed/orchestration/coordinator.py:92-405) closes a
round when enough client coroutines have pushed their ``state_dict`` through the
transport.  ``DeviceCoordinator`` offers the same calls - ``train_round``,
``start_training``, ``training_progress``, ``model_manager``, ``current_round``,
``data_dir`` and the on-disk results under ``base_dir`` - but a round is one
``FederatedEngine.run_round``: the clients are rows of a parameter arena that
train on this process's GPU (hand-written kernels where the model has them), and
aggregation is a device reduction plus one collective over the ranks.

A program written against the reference switches by swapping class names::

    coordinator = DeviceCoordinator.create(
        CoordinatorConfig(num_rounds=2, min_clients=8, min_completion_rate=1.0,
                          round_timeout=300, base_dir=Path("runs/a")),
        TrainingConfig(epochs=2, batch_size=64, learning_rate=0.1),
        model="mnist_cnn", samples_per_client=7500)
    async for metrics in coordinator.start_training():
        ...

With a ``server`` (``InNodeServer`` or ``HTTPServer``) attached, outside clients fetch
the current global model and the termination flag the way the reference's clients do.
When the engine has ``external_slots``, the updates they submit for the current round
take part in it as extra rows of the same device reduction (weight ``num_samples``, same
screening): a *mixed* round of resident and outside clients.  ``external_updates`` is the
number of outside updates a round waits for (up to ``round_timeout``; 0 = take what has
arrived).  While a server is attached the round itself runs on a worker thread of its own,
so the event loop keeps answering ``/status`` and ``/model`` during a long round.
"""

from __future__ import annotations

import asyncio
import time
from concurrent.futures import ThreadPoolExecutor
from datetime import datetime
from pathlib import Path
from typing import TYPE_CHECKING, Any, AsyncGenerator, Callable

from nanofed_amd.utils import Logger, get_current_time

from .coordinator import CoordinatorConfig
from .types import ClientInfo, RoundMetrics, RoundStatus, TrainingProgress

if TYPE_CHECKING:
    from nanofed_amd.engine import EngineConfig, FederatedEngine
    from nanofed_amd.trainer import TrainingConfig


def engine_config_from(
    config: CoordinatorConfig,
    training: "TrainingConfig | None" = None,
    **engine_kwargs: Any,
) -> "EngineConfig":
    """``EngineConfig`` that runs what ``config`` (rounds, clients, quorum, round
    time limit) and ``training`` (local epochs, batch size, learning rate) ask for;
    ``engine_kwargs`` set or override any other engine field (model, dataset,
    samples_per_client, dp, ...)."""
    from nanofed_amd.engine import EngineConfig

    fields: dict[str, Any] = dict(
        num_clients=config.min_clients,
        num_rounds=config.num_rounds,
        min_clients=config.min_clients,
        min_completion_rate=config.min_completion_rate,
        round_timeout_s=float(config.round_timeout),
    )
    if training is not None:
        fields.update(
            local_epochs=training.epochs,
            batch_size=training.batch_size,
            learning_rate=training.learning_rate,
        )
    fields.update(engine_kwargs)
    return EngineConfig(**fields)


class DeviceCoordinator:
    def __init__(
        self,
        engine: "FederatedEngine",
        config: CoordinatorConfig,
        server: Any | None = None,
        external_updates: int = 0,
        background_rounds: bool | None = None,
    ) -> None:
        self._engine = engine
        self._config = config
        self._server = server
        self._external_updates = int(external_updates)
        if self._external_updates > engine.cfg.external_slots:
            raise ValueError("external_updates exceeds the engine's external_slots")
        self._slots: dict[str, int] = {}  # outside client id -> its arena row
        # rounds off the event loop: on by default when somebody may be asking the server
        self._background = server is not None if background_rounds is None else background_rounds
        self._worker: ThreadPoolExecutor | None = None
        self._logger = Logger()
        self._clients: dict[str, ClientInfo] = {}
        self._round_metrics: list[RoundMetrics] = []
        self._status = RoundStatus.INITIALIZED
        self._round_lock = asyncio.Lock()
        self._more_rounds = False  # set by start_training: another round follows

        base = Path(config.base_dir)
        self._data_dir = base / "data"
        self._metrics_dir = base / "metrics"
        if engine.base_dir is None or Path(engine.base_dir) != base:
            # creates metrics/, data/, models/{configs,models}/ and saves the
            # bootstrap model version (rank 0 writes)
            engine.enable_persistence(base)
        if server is not None:
            server.set_coordinator(self)
            self._sync_server_round()

    @classmethod
    def create(
        cls,
        config: CoordinatorConfig,
        training: "TrainingConfig | None" = None,
        server: Any | None = None,
        comm: Any | None = None,
        data: Any | None = None,
        external_updates: int = 0,
        background_rounds: bool | None = None,
        **engine_kwargs: Any,
    ) -> "DeviceCoordinator":
        """Build the engine from the reference-style configs and wrap it."""
        from nanofed_amd.engine import FederatedEngine

        engine = FederatedEngine(
            engine_config_from(config, training, **engine_kwargs),
            comm=comm, data=data, base_dir=config.base_dir,
        )
        return cls(engine, config, server, external_updates, background_rounds)

    # -- accessors -----------------------------------------------------------
    @property
    def engine(self) -> "FederatedEngine":
        return self._engine

    @property
    def server(self) -> Any:
        return self._server

    @property
    def data_dir(self) -> Path:
        return self._data_dir

    @property
    def model_manager(self) -> Any:
        """The ``ModelManager`` that holds the saved versions (rank 0; ``None`` on
        the other ranks of a multi-GPU job, which write nothing)."""
        return self._engine._model_manager

    @property
    def current_round(self) -> int:
        return self._engine.round_id

    @property
    def round_metrics(self) -> list[RoundMetrics]:
        return list(self._round_metrics)

    @property
    def training_progress(self) -> TrainingProgress:
        return {
            "current_round": self.current_round,
            "total_rounds": self._config.num_rounds,
            "active_clients": len(self._clients),
            "global_metrics": self._get_global_metrics(),
            "status": self._status.name,
        }

    def _get_global_metrics(self) -> dict[str, float]:
        series: dict[str, list[float]] = {}
        for rm in self._round_metrics:
            for key, value in rm.agg_metrics.items():
                series.setdefault(key, []).append(value)
        return {k: sum(v) / len(v) for k, v in series.items()}

    def _sync_server_round(self) -> None:
        if self._server is None:
            return
        advance = getattr(self._server, "advance_round", None)
        if callable(advance):
            advance(self.current_round)
        else:
            self._server._current_round = self.current_round

    # -- outside clients -------------------------------------------------------
    async def _collect_external(self) -> None:
        """Move what outside clients submitted for this round into the engine's external
        rows; wait (at most ``round_timeout``) until ``external_updates`` of them are there."""
        slots = self._engine.cfg.external_slots
        server = self._server
        if not slots or server is None or not hasattr(server, "_updates"):
            return
        deadline = time.monotonic() + float(self._config.round_timeout)
        while len(server._updates) < self._external_updates and time.monotonic() < deadline:
            await asyncio.sleep(0.005)
        async with server._lock:
            updates = dict(server._updates)
            server._updates.clear()
        for cid, update in updates.items():
            slot = self._slots.get(cid)
            if slot is None:
                if len(self._slots) >= slots:
                    self._logger.warning(f"no external slot left for client {cid}; update dropped")
                    continue
                slot = self._slots[cid] = len(self._slots)
            m = dict(update.get("metrics") or {})
            n = float(m.get("num_samples") or m.get("samples_processed") or 1.0)
            self._engine.submit_external(slot, update["model_state"], n, m, client_id=cid)

    def _run_round(self):
        # inside start_training the engine queues the next round on the device while
        # this one's report is being built and handed out
        return self._engine.run_round(prefetch_next=self._more_rounds)

    def _on_worker(self):
        """The round on the coordinator's worker thread (one thread for every round, so
        HIP's current device and torch's current stream are the same from round to round)."""
        import torch

        dev = self._engine.device
        if dev.type == "cuda":
            torch.cuda.set_device(dev)
        return self._run_round()

    # -- the round -----------------------------------------------------------
    async def train_round(self) -> RoundMetrics:
        """One round on the device plane.  Like the reference's round it raises
        ``TimeoutError`` when fewer clients than the quorum delivered a usable
        update in time (the global model is then unchanged)."""
        with self._logger.context("coordinator", f"round_{self.current_round}"):
            async with self._round_lock:
                self._status = RoundStatus.IN_PROGRESS
                self._sync_server_round()
                try:
                    await self._collect_external()
                    if self._background:
                        if self._worker is None:
                            self._worker = ThreadPoolExecutor(1, thread_name_prefix="nanofed-round")
                        report = await asyncio.get_running_loop().run_in_executor(
                            self._worker, self._on_worker)
                    else:
                        report = self._run_round()
                except Exception as exc:
                    self._status = RoundStatus.FAILED
                    self._logger.error(f"Error in round {self.current_round}: {exc}")
                    raise
                now = get_current_time()
                for c in report.client_metrics:
                    if c["valid"]:
                        self._clients[c["client_id"]] = ClientInfo(
                            client_id=c["client_id"], status="submitted",
                            last_update=now, metrics=dict(c["metrics"]),
                        )
                ok = report.status == "COMPLETED"
                self._status = RoundStatus.COMPLETED if ok else RoundStatus.FAILED
                metrics = RoundMetrics(
                    round_id=report.round_id,
                    start_time=datetime.fromisoformat(report.start_time),
                    end_time=datetime.fromisoformat(report.end_time),
                    num_clients=report.num_clients,
                    agg_metrics=dict(report.agg_metrics),
                    status=self._status,
                )
                self._round_metrics.append(metrics)
                self._sync_server_round()
                await asyncio.sleep(0)  # let readers of the server run
                if not ok:
                    raise TimeoutError(
                        f"Round {report.round_id} closed with {report.num_clients} usable "
                        f"client updates, fewer than the quorum"
                    )
                return metrics

    def recover(self, failure: Exception) -> bool:
        """Roll back to the last completed round after a recoverable failure
        (needs ``checkpoint_rounds`` in the engine config).  Every rank of a
        multi-GPU job must make this call."""
        restored = self._engine._recover(failure)
        if restored:
            self._round_metrics = [
                m for m in self._round_metrics if m.round_id < self.current_round
            ]
            self._status = RoundStatus.INITIALIZED
            self._sync_server_round()
        return restored

    async def start_training(
        self,
        progress_callback: Callable[[TrainingProgress], None] | None = None,
    ) -> AsyncGenerator[RoundMetrics, None]:
        with self._logger.context("coordinator"):
            try:
                done = 0
                while done < self._config.num_rounds:
                    self._more_rounds = done + 1 < self._config.num_rounds
                    try:
                        metrics = await self.train_round()
                    except TimeoutError:
                        raise
                    except Exception as failure:  # noqa: BLE001 - the strategy decides
                        if not self.recover(failure):
                            raise
                        done = len(self._round_metrics)
                        continue
                    done += 1
                    if progress_callback:
                        progress_callback(self.training_progress)
                    yield metrics
                if self._server is not None:
                    await self._server.stop_training()
            except Exception as exc:
                self._logger.error(f"Training failed: {exc}")
                raise
            finally:
                self._more_rounds = False
                self._engine.flush()
                self._logger.info("Training completed")

    def close(self) -> None:
        if self._worker is not None:
            self._worker.shutdown(wait=True)
            self._worker = None
        self._engine.close()
