from .base import (
    BaseTrainer,
    Callback,
    TrainingConfig,
    TrainingMetrics,
)
from .callback import MetricsLogger
from .private import PrivateTrainer
from .torch import TorchTrainer

__all__ = [
    "BaseTrainer",
    "Callback",
    "TrainingConfig",
    "TrainingMetrics",
    "TorchTrainer",
    "MetricsLogger",
    "PrivateTrainer",
]
