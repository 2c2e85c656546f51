"""The federated engine of the device plane."""

from .config import DPConfig, EngineConfig
from .data import DeviceShards, FederatedData, build_federated_data
from .federation import FaultPlan, FederatedEngine, RoundReport, clients_of_rank

__all__ = [
    "DPConfig",
    "EngineConfig",
    "DeviceShards",
    "FederatedData",
    "build_federated_data",
    "FaultPlan",
    "FederatedEngine",
    "RoundReport",
    "clients_of_rank",
]
