"""Plain data carriers shared by every layer (reference: nanofed/core/types.py:11-29)."""

from __future__ import annotations

from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from typing import Any, TypedDict

import torch
from typing_extensions import NotRequired

from nanofed_amd.privacy.accountant.base import PrivacySpent


class ModelUpdate(TypedDict):
    """One client's contribution to a round.

    ``model_state`` values are tensors on the host/HTTP path and may be views
    into a device parameter arena on the in-node path.  ``privacy_spent`` is
    optional: only locally-private clients report it.
    """

    model_state: dict[str, torch.Tensor]
    client_id: str
    round_number: int
    metrics: dict[str, float]
    timestamp: datetime
    privacy_spent: NotRequired[PrivacySpent | dict[str, float] | None]


@dataclass(slots=True, frozen=True)
class ModelVersion:
    """Identity and location of one saved global model."""

    version_id: str
    timestamp: datetime
    config: dict[str, Any]
    path: Path
