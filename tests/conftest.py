"""Shared pytest configuration.

GPU tests are marked ``@pytest.mark.gpu`` and skipped automatically when no
HIP device is visible; everything else runs on the CPU (multi-process paths
use the gloo backend).
"""

from __future__ import annotations

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _quiet_logger():
    """Keep the framework logger from flooding test output."""
    from nanofed_amd.utils import LogConfig, Logger, LogLevel

    try:
        Logger().configure(LogConfig(level=LogLevel.ERROR))
    except Exception:
        pass
    yield
