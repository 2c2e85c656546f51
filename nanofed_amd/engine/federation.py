"""The federated round on the device plane.

Logical pattern of the reference (nanofed/orchestration/coordinator.py:282-382
with the HTTP client loop of examples/mnist/run_experiment.py:21-86):

    broadcast global model -> every client trains locally -> clients send
    model + metrics -> server computes n_k / sum(n) weights -> weighted sum ->
    new global model -> save version + round metrics

Here every rank (one process per GPU) hosts ``G = num_clients / world``
simulated clients in one arena pool and the round is

    1. expand-copy of the global arena into the pool          (broadcast)
    2. ``GroupTrainer.train_round``                            (local training)
    3. per-client finite / norm screening on the device        (validation -> mask)
    4. all-gather of a 6-float record per client               (metrics, counts)
    5. weights = mask * n / sum(mask * n), computed on device
    6. fused weighted sum over the local rows, then one all-reduce (FedAvg)
    7. quorum select: keep the old model if too few valid updates
    8. ONE device->host copy of the gathered records           (round report)

Steps 1-7 issue no host synchronisation.  Dropped, straggling or poisoned
clients are expressed as a zero in the participation mask, because a
collective cannot time out per peer (the reference polls for a quorum of
HTTP submissions instead, coordinator.py:205-245).

Persistence keeps the reference's formats: ``ModelManager`` versions
(``models/models/*.pt`` + ``models/configs/*.json``) and
``metrics/metrics_round_<id>.json``.
"""

from __future__ import annotations

import json
import time
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Callable

import numpy as np
import torch

from nanofed_amd import ops
from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.models import build_model
from nanofed_amd.parallel.comm import Communicator
from nanofed_amd.runtime.arena import ArenaLayout, ArenaPool, ParameterArena
from nanofed_amd.utils import Logger, get_current_time
from nanofed_amd.utils.trace import PhaseTimer

from .config import EngineConfig
from .data import DeviceShards, FederatedData, build_federated_data
from .trainers import (
    M_ACC,
    M_EPSILON,
    M_LOSS,
    M_SAMPLES,
    M_UPDATE_SQNORM,
    M_VALID,
    METRIC_COLUMNS,
    GroupTrainer,
    TorchGroupTrainer,
)


@dataclass
class RoundReport:
    round_id: int
    status: str
    num_clients: int
    agg_metrics: dict[str, float]
    client_metrics: list[dict]
    start_time: str
    end_time: str
    wall_s: float
    phases_ms: dict[str, float] = field(default_factory=dict)


@dataclass
class FaultPlan:
    """Fault injection for tests and drills: ``drop[r]`` = client ids that
    submit nothing in round r; ``poison[r]`` = client ids whose update is
    overwritten with NaNs before screening; ``delay[r]`` = {client id: seconds
    the client reports late} (left out when later than ``round_timeout_s``);
    ``fail[r]`` = message of a ``RuntimeError`` raised once in the middle of
    round r, the way a HIP or RCCL error would surface."""

    drop: dict[int, set[int]] = field(default_factory=dict)
    poison: dict[int, set[int]] = field(default_factory=dict)
    delay: dict[int, dict[int, float]] = field(default_factory=dict)
    fail: dict[int, str] = field(default_factory=dict)


def clients_of_rank(num_clients: int, world: int, rank: int) -> list[int]:
    """Contiguous block placement of clients on ranks (sizes differ by <= 1)."""
    return [int(c) for c in np.array_split(np.arange(num_clients), world)[rank]]


class FederatedEngine:
    def __init__(
        self,
        cfg: EngineConfig,
        comm: Communicator | None = None,
        data: FederatedData | None = None,
        base_dir: str | Path | None = None,
        faults: FaultPlan | None = None,
        trainer_factory: Callable[..., GroupTrainer] | None = None,
    ) -> None:
        self.cfg = cfg
        self.comm = comm or Communicator.from_env()
        self.device = self.comm.device
        self.logger = Logger()
        self.faults = faults or FaultPlan()
        self.base_dir = Path(base_dir) if base_dir is not None else None
        self.round_id = 0
        self.reports: list[RoundReport] = []

        if cfg.num_clients < self.comm.world_size:
            raise DeviceError(
                f"{cfg.num_clients} clients cannot occupy {self.comm.world_size} ranks"
            )
        self.client_ids = clients_of_rank(
            cfg.num_clients, self.comm.world_size, self.comm.rank
        )
        counts = [
            len(clients_of_rank(cfg.num_clients, self.comm.world_size, r))
            for r in range(self.comm.world_size)
        ]
        self.max_local = max(counts)
        self._rank_counts = counts

        torch.manual_seed(cfg.seed)  # identical initial model on every rank
        self._model_factory = lambda: build_model(cfg.model, **cfg.model_kwargs)
        self._host_model = self._model_factory()
        self.backend = self._resolve_backend()
        if self.backend == "native" and cfg.model == "mnist_cnn":
            from .mnist_native import make_layout

            self.layout = make_layout(self._host_model)
        else:
            self.layout = ArenaLayout.from_module(self._host_model)
            if self.device.type == "cuda":
                # convolution weights live in NHWC order on the GPU (no-op for
                # models without 4-D parameters)
                self.layout = self.layout.with_channels_last()
        self.global_arena = ParameterArena(self.layout, self.device)
        self.global_arena.load_state_dict(self._host_model.state_dict())
        self.comm.broadcast(self.global_arena.data, src=0)

        self.data = data or build_federated_data(cfg)
        self.shards = DeviceShards(
            self.data, self.client_ids, self.device, cfg.batch_size, cfg.seed,
            cfg.drop_last,
        )
        self.pool = ArenaPool(self.layout, len(self.client_ids), self.device)
        self.trainer = self._make_trainer(trainer_factory)

        K = cfg.num_clients
        self._partial = torch.zeros(self.layout.numel, dtype=torch.float32, device=self.device)
        self._local_records = torch.zeros(
            self.max_local, METRIC_COLUMNS, dtype=torch.float32, device=self.device
        )
        self._screen = torch.zeros(len(self.client_ids), 2, dtype=torch.float32, device=self.device)
        self._delta = torch.zeros(self.layout.numel, dtype=torch.float32, device=self.device)
        self._model_manager = None
        self._recovery = None
        self.recoveries = 0
        self._timer = PhaseTimer(self.device)
        if self.base_dir is not None and self.comm.is_root:
            self._setup_persistence()
        self._K = K

    # -- construction helpers ---------------------------------------------
    def _make_trainer(self, factory: Callable[..., GroupTrainer] | None) -> GroupTrainer:
        cfg = self.cfg
        if factory is not None:
            return factory(cfg, self.pool, self.shards, self._model_factory, self.data)
        if self.backend == "native" and cfg.model == "mnist_cnn":
            from .mnist_native import MnistCnnNativeTrainer

            return MnistCnnNativeTrainer(cfg, self.pool, self.shards, self.data)
        if self.backend == "native" and cfg.model.startswith("gpt2"):
            from .gpt2_native import GPT2NativeTrainer

            return GPT2NativeTrainer(cfg, self.pool, self.shards, self.data, self._host_model.cfg)
        if self.backend == "native":
            raise DeviceError(f"no native trainer for model {cfg.model!r}")
        return TorchGroupTrainer(
            cfg, self.pool, self.shards, self._model_factory,
            self.data.eval_inputs, self.data.eval_labels,
        )

    def _resolve_backend(self) -> str:
        cfg = self.cfg
        if cfg.backend != "auto":
            return cfg.backend
        on_gpu = self.device.type == "cuda"
        mnist_ok = (
            cfg.model == "mnist_cnn" and on_gpu
            and (cfg.dp is None or cfg.dp.mode == "per_sample") and cfg.batch_size == 64
        )
        gpt2_ok = False
        if cfg.model.startswith("gpt2") and on_gpu and cfg.dp is None and cfg.dtype == "bf16":
            m = self._host_model.cfg
            gpt2_ok = (
                m.n_embd == 64 * m.n_head and m.dropout == 0.0 and cfg.seq_len % 128 == 0
                and m.n_embd <= 2048
            )
        return "native" if (mnist_ok or gpt2_ok) else "torch"

    def _setup_persistence(self) -> None:
        from nanofed_amd.server.model_manager import ModelManager

        base = self.base_dir
        for sub in ("metrics", "data", "models/configs", "models/models"):
            (base / sub).mkdir(parents=True, exist_ok=True)
        self._model_manager = ModelManager(self._host_model)
        self._model_manager.set_dirs(base / "models" / "models", base / "models" / "configs")
        if self.cfg.checkpoint_rounds:
            from nanofed_amd.server.fault_tolerance import FaultTolerantCoordinator

            self._recovery = FaultTolerantCoordinator(base)

    # -- the round --------------------------------------------------------
    def run_round(self) -> RoundReport:
        cfg, comm = self.cfg, self.comm
        r = self.round_id
        t0 = time.perf_counter()
        start = get_current_time()

        timer = self._timer
        # 1. broadcast
        with timer.phase("broadcast"):
            self.pool.broadcast_from(self.global_arena.data)
        # 2. local training
        with timer.phase("local_train"):
            local = self.trainer.train_round(r)  # [G, METRIC_COLUMNS] on device
        # fault injection
        if r in self.faults.fail:
            raise RuntimeError(self.faults.fail.pop(r))
        late = {
            cid for cid, secs in self.faults.delay.get(r, {}).items()
            if cfg.round_timeout_s > 0 and secs > cfg.round_timeout_s
        }
        for k, cid in enumerate(self.client_ids):
            if cid in self.faults.poison.get(r, ()):
                self.pool[k].data.fill_(float("nan"))
            if cid in self.faults.drop.get(r, ()) or cid in late:
                local[k, M_VALID] = 0.0
        with timer.phase("screen_gather"):
            # 3. screening -> participation mask
            if cfg.validate_updates:
                self._screen_updates(local)
            # 4. gather the per-client records
            self._local_records.zero_()
            self._local_records[: len(self.client_ids)].copy_(local)
            gathered = comm.allgather(self._local_records)  # [W, max_local, M]
            records = torch.cat(
                [gathered[w, :c] for w, c in enumerate(self._rank_counts)], dim=0
            )  # [K, M] in client-id order
        with timer.phase("reduce"):
            # 5. weights on device
            mask = records[:, M_VALID]
            mass = mask * records[:, M_SAMPLES]
            total = mass.sum()
            weights = mass / total.clamp(min=1e-30)
            n_valid = mask.sum()
            ok = (n_valid >= float(cfg.required_clients())) & (total > 0)
            lo = self.client_ids[0]
            w_local = weights[lo : lo + len(self.client_ids)]
            # 6. FedAvg: fused local weighted sum + one all-reduce (rows with
            #    weight 0 are never read, so a poisoned row cannot leak NaNs)
            ops.weighted_sum(self.pool.rows(), w_local, out=self._partial)
            comm.allreduce_sum(self._partial, algo=cfg.reduce_algo)
            # 7. quorum select (device-side; the old model survives a failed round)
            self.global_arena.data.copy_(
                torch.where(ok, self._partial, self.global_arena.data)
            )
        # 8. the only host synchronisation of the round
        host = torch.cat([records.reshape(-1), ok.float().reshape(1)]).cpu()
        rec = host[:-1].view(-1, METRIC_COLUMNS).numpy()
        succeeded = bool(host[-1] > 0)
        phases = timer.collect() if cfg.phase_timing else {}

        report = self._report(r, rec, succeeded, start, time.perf_counter() - t0)
        report.phases_ms = phases
        self.reports.append(report)
        if succeeded:
            self.round_id += 1
            t_ck = time.perf_counter()
            self._persist(report)
            if self._model_manager is not None:
                report.phases_ms["checkpoint"] = (time.perf_counter() - t_ck) * 1000.0
        return report

    def _screen_updates(self, local: torch.Tensor) -> None:
        """Finite check (+ optional bound on ||theta_k - theta||) per client,
        written into the participation column without leaving the device."""
        cfg = self.cfg
        for k in range(len(self.client_ids)):
            row = self.pool[k].data
            if cfg.max_update_norm > 0:
                torch.sub(row, self.global_arena.data, out=self._delta)
                ops.sqnorm_finite(self._delta, out=self._screen[k])
            else:
                ops.sqnorm_finite(row, out=self._screen[k])
        finite = self._screen[:, 1] == 0
        valid = finite
        if cfg.max_update_norm > 0:
            valid = valid & (self._screen[:, 0] <= cfg.max_update_norm**2)
        local[:, M_UPDATE_SQNORM] = self._screen[:, 0]
        local[:, M_VALID] = local[:, M_VALID] * valid.float()

    # -- reporting / persistence ------------------------------------------
    def _report(
        self, r: int, rec: np.ndarray, ok: bool, start, wall: float
    ) -> RoundReport:
        mask = rec[:, M_VALID] > 0
        mass = rec[:, M_SAMPLES] * mask
        tot = float(mass.sum()) or 1.0
        w = mass / tot
        agg = {
            "loss": float((w * np.nan_to_num(rec[:, M_LOSS])).sum()),
            "accuracy": float((w * np.nan_to_num(rec[:, M_ACC])).sum()),
            "samples_processed": float((w * rec[:, M_SAMPLES]).sum()),
        }
        if self.cfg.dp is not None:
            agg["privacy_epsilon"] = float(rec[:, M_EPSILON].max())
        clients = [
            {
                "client_id": f"client_{cid}",
                "metrics": {
                    "loss": float(rec[cid, M_LOSS]),
                    "accuracy": float(rec[cid, M_ACC]),
                    "samples_processed": int(rec[cid, M_SAMPLES]),
                },
                "weight": float(w[cid]),
                "valid": bool(mask[cid]),
            }
            for cid in range(rec.shape[0])
        ]
        return RoundReport(
            round_id=r,
            status="COMPLETED" if ok else "FAILED",
            num_clients=int(mask.sum()),
            agg_metrics=agg,
            client_metrics=clients,
            start_time=start.isoformat(),
            end_time=get_current_time().isoformat(),
            wall_s=wall,
        )

    def _persist(self, report: RoundReport) -> None:
        if self._model_manager is None:
            return
        self._host_model.load_state_dict(self.global_state_dict())
        self._model_manager.save_model(
            config={
                "round_id": report.round_id,
                "client_metrics": report.client_metrics,
                "client_weights": {
                    c["client_id"]: c["weight"] for c in report.client_metrics
                },
                "start_time": report.start_time,
                "status": "AGGREGATING",
                "num_clients": report.num_clients,
            },
            metrics=report.agg_metrics,
        )
        path = self.base_dir / "metrics" / f"metrics_round_{report.round_id}.json"
        with path.open("w") as f:
            json.dump(
                {
                    "round_id": report.round_id,
                    "start_time": report.start_time,
                    "end_time": report.end_time,
                    "num_clients": report.num_clients,
                    "agg_metrics": report.agg_metrics,
                    "status": report.status,
                    "client_metrics": report.client_metrics,
                    "phases_ms": report.phases_ms,
                },
                f, indent=4,
            )
        if self._recovery is not None:
            from nanofed_amd.server.fault_tolerance import RoundState

            version = self._model_manager.current_version
            self._recovery.checkpoint_round(
                report.round_id,
                {},  # client models stay on the device; the global state is enough to resume
                version.version_id if version is not None else "",
                {"model_state": self.global_state_dict(), "next_round": self.round_id},
                RoundState.COMPLETED,
            )

    # -- public helpers ---------------------------------------------------
    def run(self, num_rounds: int | None = None) -> list[RoundReport]:
        """Run rounds until ``num_rounds`` more have COMPLETED or FAILED on
        quorum.  A recoverable exception inside a round (``RuntimeError`` is how
        HIP and RCCL errors surface; reference: nanofed/server/fault_tolerance.py
        :139-152) rolls the global model back to the last completed checkpoint and
        repeats from there, at most ``max_recoveries`` times."""
        out: list[RoundReport] = []
        target = num_rounds if num_rounds is not None else self.cfg.num_rounds
        while len(out) < target:
            try:
                out.append(self.run_round())
            except Exception as failure:  # noqa: BLE001 - the strategy decides
                if not self._recover(failure):
                    raise
        return out

    def _recover(self, failure: Exception) -> bool:
        """Every rank takes the same decision: rank 0 owns the checkpoints, the
        restored arena and round counter are broadcast."""
        if not self.cfg.checkpoint_rounds or self.recoveries >= self.cfg.max_recoveries:
            return False
        self.recoveries += 1
        self._timer.collect()
        flag = torch.zeros(2, dtype=torch.float32, device=self.device)
        if self.comm.is_root and self._recovery is not None:
            restored = self._recovery.handle_failure(failure, self.round_id)
            if restored is not None:
                _, state = restored
                self.global_arena.load_state_dict(state["model_state"])
                flag[0] = 1.0
                flag[1] = float(state.get("next_round", self.round_id))
            elif self._recovery.is_recoverable(failure):
                # nothing completed yet: the untouched initial model is the recovery point
                flag[0] = 1.0
                flag[1] = float(self.round_id)
        self.comm.broadcast(flag, src=0)
        if float(flag[0]) == 0.0:
            return False
        self.comm.broadcast(self.global_arena.data, src=0)
        self.round_id = int(flag[1])
        self.logger.warning(
            f"Recovered from {type(failure).__name__}: {failure}; resuming at round {self.round_id}"
        )
        return True

    def resume(self) -> int:
        """Continue a run from ``base_dir``: load the newest ModelManager version
        and take the round counter from its config (the reference always
        restarts at round 0, nanofed/server/model_manager/manager.py:144-188).
        Returns the round the engine will run next."""
        state = torch.zeros(1, dtype=torch.float32, device=self.device)
        if self.comm.is_root and self._model_manager is not None:
            versions = self._model_manager.list_versions()
            if versions:
                version = self._model_manager.load_model()
                self.global_arena.load_state_dict(self._host_model.state_dict())
                state[0] = float(version.config.get("round_id", -1) + 1)
        self.comm.broadcast(state, src=0)
        self.comm.broadcast(self.global_arena.data, src=0)
        self.round_id = int(state[0])
        return self.round_id

    def evaluate(self) -> dict[str, float]:
        return self.trainer.evaluate(self.global_arena.data)

    def global_state_dict(self) -> dict[str, torch.Tensor]:
        """CPU fp32 ``state_dict`` of the global model (checkpoint format)."""
        return self.global_arena.state_dict("cpu")

    def load_global_state_dict(self, state: dict) -> None:
        self.global_arena.load_state_dict(state)

    def synchronize(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def summary(self) -> dict:
        return {"config": asdict(self.cfg), "rounds": [asdict(r) for r in self.reports]}
