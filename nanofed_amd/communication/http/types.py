"""Wire-level message shapes (reference: nanofed/communication/http/types.py:6-50)."""

from __future__ import annotations

from typing import Any, Literal, TypedDict

from nanofed_amd.privacy.accountant import PrivacySpent

WireTensor = list[float] | list[list[float]]


class BaseResponse(TypedDict):
    status: Literal["success", "error"]
    message: str
    timestamp: str


class ClientModelUpdateRequest(TypedDict):
    client_id: str
    round_number: int
    model_state: dict[str, WireTensor]
    metrics: dict[str, float]
    timestamp: str


class ServerModelUpdateRequest(TypedDict, total=False):
    client_id: str
    round_number: int
    model_state: dict[str, Any]
    metrics: dict[str, float]
    timestamp: str
    status: Literal["success", "error"]
    message: str
    accepted: bool
    privacy_spent: PrivacySpent | dict[str, float] | None


class ModelUpdateResponse(BaseResponse):
    update_id: str
    accepted: bool


class GlobalModelResponse(BaseResponse):
    model_state: dict[str, WireTensor]
    round_number: int
    version_id: str
