"""Process-wide logger with a component context and an execution timer.

Behavioural contract (reference: nanofed/utils/logger.py:25-226):

* ``Logger()`` always returns the same object; it wraps
  ``logging.getLogger("nanofed")`` with a colour console handler.
* ``logger.context("coordinator", "round_3")`` labels the lines emitted inside
  the ``with`` block as ``coordinator.round_3``.
* ``configure(LogConfig)`` replaces the handlers (console / file / both).
* ``@log_exec`` logs ``Starting f`` / ``Completed f in X.XXs`` at DEBUG and
  ``Error in f: ...`` at ERROR, for plain and ``async`` functions alike.

Differences from the reference, on purpose: the context is stored in a
``contextvars.ContextVar`` so that concurrently running coroutines (several
simulated clients plus the coordinator in one event loop) and worker threads
no longer relabel each other's lines, and durations come from a monotonic
clock.
"""

from __future__ import annotations

import contextvars
import functools
import inspect
import logging
import sys
from contextlib import AbstractContextManager, contextmanager
from dataclasses import dataclass
from enum import Enum, auto
from pathlib import Path
from typing import Any, Callable, ClassVar, Iterator, Literal, TypeVar

from .dates import get_current_time, monotonic_s

F = TypeVar("F", bound=Callable[..., Any])

_LINE_FORMAT = "%(asctime)s - %(context)s - %(levelname)s - %(message)s"
_ROOT_CONTEXT = "nanofed"


class LogLevel(Enum):
    DEBUG = auto()
    INFO = auto()
    WARNING = auto()
    ERROR = auto()


_STD_LEVEL = {
    LogLevel.DEBUG: logging.DEBUG,
    LogLevel.INFO: logging.INFO,
    LogLevel.WARNING: logging.WARNING,
    LogLevel.ERROR: logging.ERROR,
}


@dataclass(slots=True, frozen=True)
class LogConfig:
    """Handler configuration for :meth:`Logger.configure`."""

    level: LogLevel = LogLevel.INFO
    color: bool = True
    format: str = _LINE_FORMAT
    output: Literal["console", "file", "both"] = "console"
    log_dir: Path | None = None


@dataclass(slots=True)
class LogContext:
    """``component[.subcomponent]`` label rendered into each log line."""

    _component: str
    _subcomponent: str | None = None

    def __str__(self) -> str:
        if self._subcomponent:
            return f"{self._component}.{self._subcomponent}"
        return self._component


_context_var: contextvars.ContextVar[LogContext | None] = contextvars.ContextVar(
    "nanofed_log_context", default=None
)


class Formatter(logging.Formatter):
    """Line formatter; optionally wraps the level name in ANSI colour."""

    COLORS: ClassVar[dict[str, str]] = {
        "DEBUG": "\033[36m",
        "INFO": "\033[32m",
        "WARNING": "\033[33m",
        "ERROR": "\033[31m",
        "RESET": "\033[0m",
    }

    def __init__(self, use_color: bool = True) -> None:
        super().__init__(_LINE_FORMAT)
        self.use_color = use_color

    def format(self, record: logging.LogRecord) -> str:
        if not hasattr(record, "context"):
            record.context = _ROOT_CONTEXT
        if not self.use_color:
            return super().format(record)
        plain = record.levelname
        colour = self.COLORS.get(plain)
        if colour is None:
            return super().format(record)
        record.levelname = f"{colour}{plain}{self.COLORS['RESET']}"
        try:
            return super().format(record)
        finally:
            # a record may be handed to several handlers (console + file)
            record.levelname = plain


class Logger:
    """Singleton façade over ``logging.getLogger("nanofed")``."""

    _instance: ClassVar["Logger | None"] = None
    _initialized: bool = False

    def __new__(cls) -> "Logger":
        if cls._instance is None:
            cls._instance = super().__new__(cls)
        return cls._instance

    def __init__(self) -> None:
        if self._initialized:
            return
        self._logger = logging.getLogger("nanofed")
        self._logger.setLevel(logging.DEBUG)
        if not self._logger.handlers:
            handler = logging.StreamHandler(sys.stdout)
            handler.setFormatter(Formatter(use_color=True))
            self._logger.addHandler(handler)
        self._initialized = True

    # -- context -----------------------------------------------------------
    @property
    def current_context(self) -> LogContext | None:
        return _context_var.get()

    @current_context.setter
    def current_context(self, value: LogContext | None) -> None:
        _context_var.set(value)

    @contextmanager
    def context(
        self, component: str, subcomponent: str | None = None
    ) -> Iterator["Logger"]:
        token = _context_var.set(LogContext(component, subcomponent))
        try:
            yield self
        finally:
            try:
                _context_var.reset(token)
            except ValueError:
                # An async generator may be finalised from a different task
                # context than the one that entered the block.
                _context_var.set(None)

    # -- configuration -----------------------------------------------------
    def configure(self, config: LogConfig) -> None:
        for handler in list(self._logger.handlers):
            self._logger.removeHandler(handler)
            if isinstance(handler, logging.FileHandler):
                handler.close()
        self._logger.setLevel(_STD_LEVEL[config.level])

        if config.output in ("console", "both"):
            console = logging.StreamHandler(sys.stdout)
            console.setFormatter(Formatter(config.color))
            self._logger.addHandler(console)

        if config.output in ("file", "both") and config.log_dir:
            stamp = get_current_time().strftime("%Y%m%d_%H%M%S")
            path = Path(config.log_dir) / f"nanofed_{stamp}.log"
            path.parent.mkdir(parents=True, exist_ok=True)
            to_file = logging.FileHandler(path)
            to_file.setFormatter(Formatter(use_color=False))
            self._logger.addHandler(to_file)

    # -- emitters ----------------------------------------------------------
    def _log(self, level: int, msg: str) -> None:
        ctx = _context_var.get()
        self._logger.log(
            level, msg, extra={"context": str(ctx) if ctx else _ROOT_CONTEXT}
        )

    def debug(self, msg: str) -> None:
        self._log(logging.DEBUG, msg)

    def info(self, msg: str) -> None:
        self._log(logging.INFO, msg)

    def warning(self, msg: str) -> None:
        self._log(logging.WARNING, msg)

    def error(self, msg: str) -> None:
        self._log(logging.ERROR, msg)


class LoggerContextManager(AbstractContextManager):
    """Object form of :meth:`Logger.context` (kept for API parity)."""

    __slots__ = ("_logger", "_context", "_token")

    def __init__(self, logger: Logger, context: LogContext) -> None:
        self._logger = logger
        self._context = context
        self._token: contextvars.Token | None = None

    def __enter__(self) -> Logger:
        self._token = _context_var.set(self._context)
        return self._logger

    def __exit__(self, *exc: Any) -> None:
        if self._token is not None:
            _context_var.reset(self._token)
            self._token = None


def log_exec(func: F) -> F:
    """Log start, duration and failure of ``func`` (sync or coroutine)."""

    name = func.__name__

    def _done(t0: float) -> None:
        Logger().debug(f"Completed {name} in {monotonic_s() - t0:.2f}s")

    if inspect.iscoroutinefunction(func):

        @functools.wraps(func)
        async def timed_async(*args: Any, **kwargs: Any) -> Any:
            log = Logger()
            t0 = monotonic_s()
            log.debug(f"Starting {name}")
            try:
                out = await func(*args, **kwargs)
            except Exception as exc:
                log.error(f"Error in {name}: {exc}")
                raise
            _done(t0)
            return out

        return timed_async  # type: ignore[return-value]

    @functools.wraps(func)
    def timed(*args: Any, **kwargs: Any) -> Any:
        log = Logger()
        t0 = monotonic_s()
        log.debug(f"Starting {name}")
        try:
            out = func(*args, **kwargs)
        except Exception as exc:
            log.error(f"Error in {name}: {exc}")
            raise
        _done(t0)
        return out

    return timed  # type: ignore[return-value]
