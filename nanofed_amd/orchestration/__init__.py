from .coordinator import Coordinator, CoordinatorConfig
from .device import DeviceCoordinator, engine_config_from
from .types import ClientInfo, RoundMetrics, RoundStatus, TrainingProgress
from .utils import coordinate

__all__ = [
    "Coordinator",
    "CoordinatorConfig",
    "DeviceCoordinator",
    "engine_config_from",
    "ClientInfo",
    "RoundMetrics",
    "RoundStatus",
    "TrainingProgress",
    "coordinate",
]
