"""Gaussian and Laplacian samplers
(reference: nanofed/privacy/noise/generators.py:14-67).

The host classes draw from a CPU ``torch.Generator``.  The device classes
(:mod:`nanofed_amd.privacy.noise.device`) draw from a counter-based Philox
stream inside a HIP kernel; they share the argument validation below.
"""

from __future__ import annotations

import functools
from typing import Any, Callable, TypeVar

import torch

from ..exceptions import NoiseGenerationError
from ..types import Shape, Tensor
from .base import BaseNoiseGenerator

G = TypeVar("G", bound=Callable[..., torch.Tensor])


def _check_shape_and_scale(shape: Any, scale: Any) -> None:
    if not shape:
        raise ValueError("Shape must be provided")
    if not isinstance(shape, tuple):
        raise ValueError("Shape must be a tuple")
    for dim in shape:
        if not isinstance(dim, int) or isinstance(dim, bool) or dim <= 0:
            raise ValueError(
                "Invalid shape: must be a tuple of positive integers"
            )
    if isinstance(scale, bool) or not isinstance(scale, (int, float)):
        raise ValueError("Scale must be a number")
    if scale <= 0:
        raise ValueError("Scale must be positive")


def validate_noise_input(func: G) -> G:
    """Reject malformed ``(shape, scale)`` with ``ValueError``; wrap anything
    the sampler itself raises in ``NoiseGenerationError``."""

    @functools.wraps(func)
    def checked(self: Any, *args: Any, **kwargs: Any) -> torch.Tensor:
        shape = args[0] if len(args) > 0 else kwargs.get("shape")
        scale = args[1] if len(args) > 1 else kwargs.get("scale")
        _check_shape_and_scale(shape, scale)
        try:
            return func(self, *args, **kwargs)
        except Exception as exc:
            raise NoiseGenerationError(
                f"Noise generation failed: {exc}"
            ) from exc

    return checked  # type: ignore[return-value]


class GaussianNoiseGenerator(BaseNoiseGenerator):
    """N(0, scale^2) samples."""

    @validate_noise_input
    def generate(self, shape: Shape, scale: float) -> Tensor:
        return torch.randn(shape, generator=self._rng).mul_(scale)


class LaplacianNoiseGenerator(BaseNoiseGenerator):
    """Laplace(0, scale) samples by inverse CDF.

    With u ~ U[0,1) and v = u - 1/2 the reference returns
    ``sign(v) * scale * log1p(-2|v|)`` (no leading minus).  The Laplace
    distribution is symmetric, so this is the same distribution as the
    textbook form; the sign convention is kept so that seeded streams match.
    """

    @validate_noise_input
    def generate(self, shape: Shape, scale: float) -> Tensor:
        v = torch.rand(shape, generator=self._rng).sub_(0.5)
        return torch.sign(v) * scale * torch.log1p(-2.0 * v.abs())
