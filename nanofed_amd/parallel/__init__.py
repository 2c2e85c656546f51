"""Multi-GPU transport: one process per GPU over RCCL/xGMI (or gloo on CPU)."""

from .comm import RS_AG_THRESHOLD_BYTES, Communicator

__all__ = ["Communicator", "RS_AG_THRESHOLD_BYTES"]
