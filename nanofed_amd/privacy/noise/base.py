"""Noise generator interface (reference: nanofed/privacy/noise/base.py:9-31)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Protocol

import torch

from ..types import Shape, Tensor


class NoiseGenerator(Protocol):
    def generate(self, shape: Shape, scale: float) -> Tensor: ...
    def set_seed(self, seed: int) -> None: ...


class BaseNoiseGenerator(ABC):
    """Host generator backed by a CPU ``torch.Generator``.

    Two generators given the same seed produce identical streams.
    """

    def __init__(self, seed: int | None = None) -> None:
        self._rng = torch.Generator()
        if seed is not None:
            self.set_seed(seed)

    def set_seed(self, seed: int) -> None:
        self._rng.manual_seed(seed)

    @abstractmethod
    def generate(self, shape: Shape, scale: float) -> Tensor: ...
