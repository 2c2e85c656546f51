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


def pytest_runtest_logreport(report):
    """Keep the full text of every failure in a file as well: a failure that does not
    come back on the next run (multi-process tests share ports and CPUs with whatever else
    the machine is doing) can still be read afterwards."""
    if report.failed:
        path = os.environ.get("NANOFED_TEST_FAILURE_LOG", "/tmp/nanofed_test_failures.log")
        try:
            with open(path, "a") as fh:
                fh.write(f"==== {report.nodeid} [{report.when}]\n{report.longreprtext}\n")
                for name, text in report.sections:
                    fh.write(f"---- {name}\n{text[-4000:]}\n")
        except OSError:
            pass


@pytest.fixture(autouse=True)
def _quiet_logger():
    """Keep the framework logger from flooding test output."""
    from nanofed_amd.utils import LogConfig, Logger, LogLevel

    try:
        Logger().configure(LogConfig(level=LogLevel.ERROR))
    except Exception:
        pass
    yield
