from .base import BaseNoiseGenerator, NoiseGenerator
from .generators import (
    GaussianNoiseGenerator,
    LaplacianNoiseGenerator,
    validate_noise_input,
)

__all__ = [
    "NoiseGenerator",
    "BaseNoiseGenerator",
    "GaussianNoiseGenerator",
    "LaplacianNoiseGenerator",
    "validate_noise_input",
    "DeviceGaussianNoiseGenerator",
    "DeviceLaplacianNoiseGenerator",
]


def __getattr__(name: str):
    # The device generators pull in the HIP op layer; import them lazily so
    # that the host-only privacy API has no dependency on it.
    if name in ("DeviceGaussianNoiseGenerator", "DeviceLaplacianNoiseGenerator"):
        from . import device

        return getattr(device, name)
    raise AttributeError(name)
