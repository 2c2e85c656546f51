"""Structural interfaces between layers (reference: nanofed/core/interfaces.py:13-67).

``AggregatorProtoocol`` keeps the reference's public (misspelled) name;
``AggregatorProtocol`` is the corrected alias for the same object.
"""

from __future__ import annotations

from pathlib import Path
from typing import Any, Iterator, Protocol, TypeVar, runtime_checkable

import torch
from torch import nn
from torch.utils.data import DataLoader

from .types import ModelVersion

T = TypeVar("T", bound=nn.Module)


@runtime_checkable
class ModelProtocol(Protocol):
    def forward(self, x: torch.Tensor) -> torch.Tensor: ...
    def parameters(self) -> Iterator[torch.nn.Parameter]: ...
    def state_dict(self) -> dict[str, torch.Tensor]: ...
    def load_state_dict(self, state_dict: dict[str, torch.Tensor]) -> Any: ...
    def to(self, device: str | torch.device) -> "ModelProtocol": ...


class AggregatorProtoocol(Protocol[T]):
    def aggregate(self, updates: list[T]) -> T: ...


AggregatorProtocol = AggregatorProtoocol


class TrainerProtocol(Protocol[T]):
    def train(self, model: T, data: DataLoader) -> T: ...
    def validate(self, model: T, data: DataLoader) -> dict[str, float]: ...


class ModelManagerProtocol(Protocol):
    def set_dirs(self, models_dir: Path, configs_dir: Path) -> None: ...
    @property
    def current_version(self) -> Any: ...
    def load_model(self) -> Any: ...
    def save_model(
        self, config: dict[str, Any], metrics: dict[str, float] | None
    ) -> Any: ...
    def list_versions(self) -> list[ModelVersion]: ...
    @property
    def model(self) -> ModelProtocol: ...


class CoordinatorProtocol(Protocol):
    @property
    def model_manager(self) -> ModelManagerProtocol: ...


class ServerProtocol(Protocol):
    @property
    def host(self) -> str: ...
    @property
    def port(self) -> int: ...
    @property
    def url(self) -> str: ...
