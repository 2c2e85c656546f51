"""Noise generators that draw on the device (reference API: nanofed/privacy/noise/
generators.py:49-67, which draws on the host from a CPU ``torch.Generator``).

Samples come from the counter-based Philox4x32-10 stream of ``csrc/common/philox.h``
(``ops.philox_fill``): sample i of call c of seed s is a pure function of (s, c, i), so

* the same seed gives the same samples on the GPU and on the CPU (the host evaluates the
  same stream, ``csrc/host/philox_cpu.cpp``), and two generators with one seed agree -
  the reproducibility contract of the host classes;
* nothing is drawn on the host and copied: a DP step that perturbs a model on the GPU
  launches one fill kernel per tensor instead of ``randn`` + a host-to-device copy.

The streams are not those of ``torch.randn`` (a different generator), which is why the
host classes stay the default wherever a seeded host stream may be expected.
"""

from __future__ import annotations

import os

import torch

from ..types import Shape, Tensor
from .generators import validate_noise_input


class _DeviceNoiseGenerator:
    _MODE = 1

    def __init__(self, seed: int | None = None, device: str | torch.device | None = None) -> None:
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self._device = torch.device(device)
        self._seed = int.from_bytes(os.urandom(7), "little") if seed is None else int(seed)
        self._calls = 0

    @property
    def device(self) -> torch.device:
        return self._device

    def set_seed(self, seed: int) -> None:
        self._seed = int(seed)
        self._calls = 0

    def _fill(self, shape: Shape, scale: float) -> Tensor:
        from nanofed_amd import ops

        out = torch.empty(tuple(shape), dtype=torch.float32, device=self._device)
        ops.philox_fill(out, self._seed, self._calls, mode=self._MODE, scale=float(scale))
        self._calls += 1
        return out


class DeviceGaussianNoiseGenerator(_DeviceNoiseGenerator):
    """N(0, scale^2) samples, drawn where they are used."""

    _MODE = 1  # ops.MODE_NORMAL

    @validate_noise_input
    def generate(self, shape: Shape, scale: float) -> Tensor:
        return self._fill(shape, scale)


class DeviceLaplacianNoiseGenerator(_DeviceNoiseGenerator):
    """Laplace(0, scale) samples (inverse CDF of a Philox uniform), drawn where they are
    used."""

    _MODE = 2  # ops.MODE_LAPLACE

    @validate_noise_input
    def generate(self, shape: Shape, scale: float) -> Tensor:
        return self._fill(shape, scale)
