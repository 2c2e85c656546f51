from .aggregator import AggregationResult, BaseAggregator, FedAvgAggregator
from .fault_tolerance import (
    CheckpointMetadata,
    FaultTolerantCoordinator,
    FileStateStore,
    RoundState,
    SimpleRecoveryStrategy,
)
from .model_manager import ModelManager, ModelVersion

__all__ = [
    "AggregationResult",
    "BaseAggregator",
    "CheckpointMetadata",
    "FaultTolerantCoordinator",
    "FedAvgAggregator",
    "FileStateStore",
    "ModelManager",
    "ModelVersion",
    "RoundState",
    "SimpleRecoveryStrategy",
]
