"""Wall-clock helpers.

Every timestamp that ends up in metrics files, checkpoints or model version
ids is a timezone-aware UTC ``datetime`` (reference: nanofed/utils/dates.py:4-5).
Durations, in contrast, are measured with a monotonic clock so that an NTP
step during a round cannot produce a negative "Completed in" figure.
"""

from __future__ import annotations

import time
from datetime import datetime, timezone


def get_current_time() -> datetime:
    """Current time as an aware UTC datetime."""
    return datetime.now(tz=timezone.utc)


def monotonic_s() -> float:
    """Monotonic seconds, for measuring durations only."""
    return time.perf_counter()
