from .config import NoiseType, PrivacyConfig
from .constants import (
    DEFAULT_DELTA,
    DEFAULT_EPSILON,
    DEFAULT_MAX_GRAD_NORM,
    DEFAULT_NOISE_MULTIPLIER,
)
from .exceptions import (
    NoiseGenerationError,
    PrivacyBudgetExceededError,
    PrivacyConfigurationError,
    PrivacyError,
)
from .noise import GaussianNoiseGenerator, LaplacianNoiseGenerator

__all__ = [
    "NoiseType",
    "PrivacyConfig",
    "DEFAULT_DELTA",
    "DEFAULT_EPSILON",
    "DEFAULT_NOISE_MULTIPLIER",
    "DEFAULT_MAX_GRAD_NORM",
    "PrivacyError",
    "PrivacyBudgetExceededError",
    "PrivacyConfigurationError",
    "NoiseGenerationError",
    "GaussianNoiseGenerator",
    "LaplacianNoiseGenerator",
]
