"""MI355X-native federated learning with the NanoFed API surface.

Top-level names match ``nanofed/__init__.py:12-29`` of the reference; the
device-plane additions live in ``nanofed_amd.engine``, ``nanofed_amd.ops``,
``nanofed_amd.parallel`` and ``nanofed_amd.runtime``.
"""

from nanofed_amd.communication import HTTPClient, HTTPServer
from nanofed_amd.orchestration import (
    Coordinator,
    CoordinatorConfig,
    DeviceCoordinator,
    coordinate,
)
from nanofed_amd.server import FedAvgAggregator, ModelManager
from nanofed_amd.trainer import PrivateTrainer, TorchTrainer, TrainingConfig

__version__ = "0.2.0"

__all__ = [
    "HTTPClient",
    "HTTPServer",
    "TrainingConfig",
    "TorchTrainer",
    "PrivateTrainer",
    "Coordinator",
    "CoordinatorConfig",
    "DeviceCoordinator",
    "FedAvgAggregator",
    "ModelManager",
    "coordinate",
    "__version__",
]
