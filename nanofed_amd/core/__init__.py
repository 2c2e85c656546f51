from .exceptions import (
    AggregationError,
    DeviceError,
    ModelManagerError,
    NanoFedError,
)
from .interfaces import (
    AggregatorProtocol,
    AggregatorProtoocol,
    CoordinatorProtocol,
    ModelManagerProtocol,
    ModelProtocol,
    ServerProtocol,
    TrainerProtocol,
)
from .types import ModelUpdate, ModelVersion

__all__ = [
    "NanoFedError",
    "AggregationError",
    "ModelManagerError",
    "DeviceError",
    "ModelProtocol",
    "AggregatorProtoocol",
    "AggregatorProtocol",
    "TrainerProtocol",
    "ServerProtocol",
    "ModelManagerProtocol",
    "CoordinatorProtocol",
    "ModelUpdate",
    "ModelVersion",
]
