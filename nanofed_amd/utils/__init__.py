from .dates import get_current_time, monotonic_s
from .logger import LogConfig, Logger, LogLevel, log_exec

__all__ = [
    "Logger",
    "log_exec",
    "LogLevel",
    "LogConfig",
    "get_current_time",
    "monotonic_s",
]
