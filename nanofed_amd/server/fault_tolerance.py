"""Round checkpointing and failure recovery
(reference: nanofed/server/fault_tolerance.py:14-212).

On-disk contract::

    <base_dir>/checkpoints/round_<id>/metadata.json   CheckpointMetadata.to_dict()
    <base_dir>/checkpoints/round_<id>/state.pt        torch.save(state dict)

``metadata.json`` stores client updates with list-valued ``model_state`` (tensors
are converted on the way out, restored to tensors on the way in) and ISO
timestamps, so the file is plain JSON.  Recoverable failures are
``TimeoutError``, ``ConnectionError`` and ``RuntimeError`` (HIP and RCCL
errors surface as ``RuntimeError``); the recovery point is the COMPLETED
checkpoint with the greatest round id.  Writes go to a temporary name first and
are renamed, so a crash mid-write never leaves a half-written checkpoint to be
picked as the recovery point.
"""

from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass
from datetime import datetime
from enum import Enum, auto
from pathlib import Path
from typing import Any, Protocol

import torch

from nanofed_amd.core import ModelUpdate
from nanofed_amd.utils import Logger, get_current_time


class RoundState(Enum):
    INITIALIZED = auto()
    IN_PROGRESS = auto()
    FAILED = auto()
    COMPLETED = auto()


def _jsonable(value: Any) -> Any:
    if isinstance(value, torch.Tensor):
        return value.detach().cpu().tolist()
    if isinstance(value, dict):
        return {str(k): _jsonable(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_jsonable(v) for v in value]
    if isinstance(value, datetime):
        return value.isoformat()
    if isinstance(value, Enum):
        return value.name
    if hasattr(value, "__dataclass_fields__"):
        return {f: _jsonable(getattr(value, f)) for f in value.__dataclass_fields__}
    return value


@dataclass(slots=True, frozen=True)
class CheckpointMetadata:
    round_id: int
    timestamp: datetime
    num_clients: int
    client_updates: dict[str, ModelUpdate]
    global_model_version: str
    state: RoundState

    def to_dict(self) -> dict[str, Any]:
        return {
            "round_id": self.round_id,
            "timestamp": self.timestamp.isoformat(),
            "num_clients": self.num_clients,
            "client_updates": _jsonable(self.client_updates),
            "global_model_version": self.global_model_version,
            "state": self.state.name,
        }

    @staticmethod
    def from_dict(data: dict[str, Any]) -> "CheckpointMetadata":
        updates = data["client_updates"]
        for update in updates.values():
            state = update.get("model_state", {})
            for name, value in state.items():
                if not isinstance(value, torch.Tensor):
                    state[name] = torch.tensor(value)
        return CheckpointMetadata(
            round_id=data["round_id"],
            timestamp=datetime.fromisoformat(data["timestamp"]),
            num_clients=data["num_clients"],
            client_updates=updates,
            global_model_version=data["global_model_version"],
            state=RoundState[data["state"]],
        )


class StateStore(Protocol):
    def save_checkpoint(
        self, metadata: CheckpointMetadata, state: dict[str, Any]
    ) -> None: ...

    def load_checkpoint(
        self, round_id: int
    ) -> tuple[CheckpointMetadata, dict[str, Any]] | None: ...

    def list_checkpoints(self) -> list[CheckpointMetadata]: ...


class RecoveryStrategy(Protocol):
    def should_recover(self, failure: Exception) -> bool: ...

    def get_recovery_point(
        self, checkpoints: list[CheckpointMetadata]
    ) -> CheckpointMetadata | None: ...


def _cpu_tree(value: Any) -> Any:
    if isinstance(value, torch.Tensor):
        return value.detach().to("cpu")
    if isinstance(value, dict):
        return {k: _cpu_tree(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return type(value)(_cpu_tree(v) for v in value)
    return value


class FileStateStore(StateStore):
    def __init__(self, base_dir: Path) -> None:
        self._base_dir = Path(base_dir) / "checkpoints"
        self._base_dir.mkdir(parents=True, exist_ok=True)
        self._logger = Logger()

    def _round_dir(self, round_id: int) -> Path:
        return self._base_dir / f"round_{round_id}"

    def save_checkpoint(
        self, metadata: CheckpointMetadata, state: dict[str, Any]
    ) -> None:
        target = self._round_dir(metadata.round_id)
        target.mkdir(exist_ok=True)
        tmp_state = target / "state.pt.tmp"
        torch.save(_cpu_tree(state), tmp_state)
        os.replace(tmp_state, target / "state.pt")
        # metadata last: its presence marks the checkpoint as complete
        tmp_meta = target / "metadata.json.tmp"
        with open(tmp_meta, "w") as f:
            json.dump(metadata.to_dict(), f)
        os.replace(tmp_meta, target / "metadata.json")
        self._logger.info(f"Saved checkpoint for round {metadata.round_id}")

    def load_checkpoint(
        self, round_id: int
    ) -> tuple[CheckpointMetadata, dict[str, Any]] | None:
        source = self._round_dir(round_id)
        if not (source / "metadata.json").exists():
            return None
        with open(source / "metadata.json") as f:
            metadata = CheckpointMetadata.from_dict(json.load(f))
        state = torch.load(source / "state.pt", weights_only=True)
        self._logger.info(f"Loaded checkpoint for round {round_id}")
        return metadata, state

    def list_checkpoints(self) -> list[CheckpointMetadata]:
        def round_no(path: Path) -> int:
            m = re.search(r"round_(\d+)$", path.name)
            return int(m.group(1)) if m else -1

        found = []
        for path in sorted(self._base_dir.glob("round_*"), key=round_no):
            meta = path / "metadata.json"
            if meta.exists():
                with open(meta) as f:
                    found.append(CheckpointMetadata.from_dict(json.load(f)))
        return found


class SimpleRecoveryStrategy(RecoveryStrategy):
    RECOVERABLE = (TimeoutError, ConnectionError, RuntimeError)

    def should_recover(self, failure: Exception) -> bool:
        return isinstance(failure, self.RECOVERABLE)

    def get_recovery_point(
        self, checkpoints: list[CheckpointMetadata]
    ) -> CheckpointMetadata | None:
        done = [c for c in checkpoints if c.state == RoundState.COMPLETED]
        return max(done, key=lambda c: c.round_id) if done else None


class FaultTolerantCoordinator:
    def __init__(
        self,
        base_dir: Path,
        state_store: StateStore | None = None,
        recovery_strategy: RecoveryStrategy | None = None,
    ) -> None:
        self._state_store = state_store or FileStateStore(base_dir)
        self._recovery = recovery_strategy or SimpleRecoveryStrategy()
        self._logger = Logger()

    def checkpoint_round(
        self,
        round_id: int,
        client_updates: dict[str, ModelUpdate],
        model_version: str,
        state: dict[str, Any],
        round_state: RoundState,
    ) -> None:
        self._state_store.save_checkpoint(
            CheckpointMetadata(
                round_id=round_id,
                timestamp=get_current_time(),
                num_clients=len(client_updates),
                client_updates=client_updates,
                global_model_version=model_version,
                state=round_state,
            ),
            state,
        )

    def restore_round(
        self, round_id: int
    ) -> tuple[CheckpointMetadata, dict[str, Any]] | None:
        return self._state_store.load_checkpoint(round_id)

    def is_recoverable(self, failure: Exception) -> bool:
        """The recovery strategy's verdict on ``failure``."""
        return self._recovery.should_recover(failure)

    def handle_failure(
        self, failure: Exception, current_round: int
    ) -> tuple[CheckpointMetadata, dict[str, Any]] | None:
        """Latest COMPLETED checkpoint for a recoverable failure, else ``None``."""
        if not self._recovery.should_recover(failure):
            self._logger.error(
                f"Unrecoverable failure in round {current_round}: {failure}"
            )
            return None
        point = self._recovery.get_recovery_point(
            self._state_store.list_checkpoints()
        )
        if point is None:
            self._logger.error("No valid recovery point found")
            return None
        self._logger.info(f"Recovering from round {point.round_id}")
        return self.restore_round(point.round_id)
