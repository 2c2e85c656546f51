"""Versioned storage of the global model
(reference: nanofed/server/model_manager/manager.py:13-210).

On-disk contract::

    <models_dir>/model_v_<YYYYmmdd_HHMMSS>_<NNN>.pt    torch.save(state_dict), CPU fp32
    <configs_dir>/model_v_<YYYYmmdd_HHMMSS>_<NNN>.json {"version_id", "timestamp", "config"}

The ``.pt`` always holds plain CPU tensors keyed by the model's state-dict
names, whatever the in-memory representation (device arena, bf16 shadows).
"Latest" means greatest (timestamp, counter) parsed from the version id, not
the lexicographically last file name.  ``metrics`` passed to ``save_model``
are written into the JSON next to ``config`` (the reference accepts and drops
them).
"""

from __future__ import annotations

import json
import re
from dataclasses import asdict, is_dataclass
from datetime import datetime
from pathlib import Path
from typing import Any

import torch

from nanofed_amd.core import ModelManagerError, ModelProtocol, ModelVersion
from nanofed_amd.utils import Logger, get_current_time, log_exec

_VERSION_RE = re.compile(r"model_v_(\d{8})_(\d{6})_(\d+)$")


def make_json_serializable(data: Any) -> Any:
    """Recursively coerce ``data`` into JSON types; unknown objects -> ``str``."""
    if isinstance(data, dict):
        return {str(k): make_json_serializable(v) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return [make_json_serializable(v) for v in data]
    if is_dataclass(data) and not isinstance(data, type):
        return make_json_serializable(asdict(data))
    if data is None or isinstance(data, (bool, int, float, str)):
        return data
    if isinstance(data, torch.Tensor) and data.numel() == 1:
        return data.item()
    return str(data)


def _version_sort_key(version_id: str) -> tuple[str, str, int, str]:
    m = _VERSION_RE.search(version_id)
    if not m:
        return ("", "", -1, version_id)
    return (m.group(1), m.group(2), int(m.group(3)), version_id)


def _owns_storage(t: torch.Tensor) -> bool:
    """True when ``t`` is a contiguous CPU tensor that fills its storage: saving it writes
    exactly its bytes (a view into a larger buffer would drag the whole buffer along)."""
    return (t.device.type == "cpu" and t.is_contiguous() and t.storage_offset() == 0
            and t.untyped_storage().nbytes() == t.numel() * t.element_size())


def _cpu_state(model: ModelProtocol) -> dict[str, torch.Tensor]:
    """The state to save: detached host tensors, each with a storage of its own (copied
    only where that is not already the case)."""
    out = {}
    for k, v in model.state_dict().items():
        if isinstance(v, torch.Tensor):
            v = v.detach()
            v = v if _owns_storage(v) else v.to("cpu").contiguous().clone()
        out[k] = v
    return out


class ModelManager:
    def __init__(self, model: ModelProtocol) -> None:
        self._model = model
        self._logger = Logger()
        self._current_version: ModelVersion | None = None
        self._version_counter: int = 0
        self._models_dir: Path | None = None
        self._configs_dir: Path | None = None

    def set_dirs(self, models_dir: Path, configs_dir: Path) -> None:
        """Choose the storage directories; on an empty store the current
        weights are saved as the bootstrap version."""
        self._models_dir = Path(models_dir)
        self._configs_dir = Path(configs_dir)
        if not self.list_versions():
            self._logger.info("No model versions found. Saving initial model.")
            self.save_model(config={"name": "default", "version": "1.0"})

    @property
    def current_version(self) -> ModelVersion | None:
        return self._current_version

    @property
    def model(self) -> ModelProtocol:
        return self._model

    def _dirs(self) -> tuple[Path, Path]:
        if not self._models_dir or not self._configs_dir:
            raise ModelManagerError("Directories not set. Call set_dirs first.")
        return self._models_dir, self._configs_dir

    def _generate_version_id(self) -> str:
        self._version_counter += 1
        stamp = get_current_time().strftime("%Y%m%d_%H%M%S")
        return f"model_v_{stamp}_{self._version_counter:03d}"

    @log_exec
    def save_model(
        self, config: dict[str, Any], metrics: dict[str, float] | None = None
    ) -> ModelVersion:
        models_dir, configs_dir = self._dirs()
        with self._logger.context("model_manager", "save"):
            version_id = self._generate_version_id()
            model_path = models_dir / f"{version_id}.pt"
            torch.save(_cpu_state(self._model), model_path)

            now = get_current_time()
            record: dict[str, Any] = {
                "version_id": version_id,
                "timestamp": now.isoformat(),
                "config": make_json_serializable(config),
            }
            if metrics is not None:
                record["metrics"] = make_json_serializable(metrics)
            try:
                with open(configs_dir / f"{version_id}.json", "w") as f:
                    json.dump(record, f, indent=2)
            except TypeError as exc:
                raise ModelManagerError(
                    f"Failed to serialize config data: {exc}"
                ) from exc

            version = ModelVersion(
                version_id=version_id, timestamp=now, config=config,
                path=model_path,
            )
            self._current_version = version
            self._logger.info(f"Saved model version: {version_id}")
            return version

    def _config_paths(self) -> list[Path]:
        _, configs_dir = self._dirs()
        return sorted(
            configs_dir.glob("*.json"), key=lambda p: _version_sort_key(p.stem)
        )

    @staticmethod
    def _read_version(config_path: Path, models_dir: Path) -> ModelVersion:
        with open(config_path) as f:
            data = json.load(f)
        return ModelVersion(
            version_id=data["version_id"],
            timestamp=datetime.fromisoformat(data["timestamp"]),
            config=data["config"],
            path=models_dir / f"{data['version_id']}.pt",
        )

    @log_exec
    def load_model(self, version_id: str | None = None) -> ModelVersion:
        models_dir, configs_dir = self._dirs()
        with self._logger.context("model_manager", "load"):
            if version_id is None:
                candidates = self._config_paths()
                if not candidates:
                    raise ModelManagerError("No model versions found")
                config_path = candidates[-1]
            else:
                config_path = configs_dir / f"{version_id}.json"
                if not config_path.exists():
                    raise ModelManagerError(f"Version {version_id} not found")

            version = self._read_version(config_path, models_dir)
            if not version.path.exists():
                raise ModelManagerError(
                    f"Model file not found for version {version.version_id}"
                )
            try:
                state = torch.load(version.path, weights_only=True)
                self._model.load_state_dict(state)
            except Exception as exc:
                raise ModelManagerError(f"Failed to load model: {exc}") from exc

            self._current_version = version
            self._logger.info(f"Loaded model version: {version.version_id}")
            return version

    def list_versions(self) -> list[ModelVersion]:
        models_dir, _ = self._dirs()
        return [self._read_version(p, models_dir) for p in self._config_paths()]
