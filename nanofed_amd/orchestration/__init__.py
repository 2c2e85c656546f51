from .coordinator import Coordinator, CoordinatorConfig
from .types import ClientInfo, RoundMetrics, RoundStatus, TrainingProgress
from .utils import coordinate

__all__ = [
    "Coordinator",
    "CoordinatorConfig",
    "ClientInfo",
    "RoundMetrics",
    "RoundStatus",
    "TrainingProgress",
    "coordinate",
]
