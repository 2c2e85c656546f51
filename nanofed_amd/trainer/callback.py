"""JSON metrics logger callback (reference: nanofed/trainer/callback.py:9-53)."""

from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path

from nanofed_amd.utils import get_current_time

from .base import Callback, TrainingMetrics


@dataclass(slots=True)
class MetricsLogger(Callback):
    """Buffers one record per batch and per epoch; rewrites
    ``<log_dir>/<experiment_name>_<YYYYmmdd_HHMMSS>.json`` at every epoch end."""

    log_dir: Path
    experiment_name: str

    def __post_init__(self) -> None:
        self.log_dir.mkdir(parents=True, exist_ok=True)
        stamp = get_current_time().strftime("%Y%m%d_%H%M%S")
        self._log_file = self.log_dir / f"{self.experiment_name}_{stamp}.json"
        self._metrics: list[dict] = []

    def _record(self, kind: str, metrics: TrainingMetrics, **extra: int) -> dict:
        return {
            "type": kind,
            **extra,
            "loss": metrics.loss,
            "accuracy": metrics.accuracy,
            "samples_processed": metrics.samples_processed,
            "timestamp": get_current_time().isoformat(),
        }

    def on_eopch_start(self, epoch: int) -> None:
        return None

    on_epoch_start = on_eopch_start

    def on_epoch_end(self, epoch: int, metrics: TrainingMetrics) -> None:
        self._metrics.append(self._record("epoch", metrics, epoch=epoch))
        with open(self._log_file, "w") as f:
            json.dump(self._metrics, f, indent=2)

    def on_batch_end(self, batch: int, metrics: TrainingMetrics) -> None:
        self._metrics.append(
            self._record("batch", metrics, epoch=metrics.epoch, batch=batch)
        )
