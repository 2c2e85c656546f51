"""Framework exception hierarchy (reference: nanofed/core/exceptions.py:1-16)."""


class NanoFedError(Exception):
    """Root of all framework errors raised outside the privacy package."""


class AggregationError(NanoFedError):
    """Client updates cannot be combined (empty, mixed rounds, mixed keys)."""


class ModelManagerError(NanoFedError):
    """Model version storage / retrieval failed."""


class DeviceError(NanoFedError):
    """A device-plane operation (HIP kernel, arena, RCCL collective) failed or
    was invoked with operands that do not match what the kernel assumes."""
