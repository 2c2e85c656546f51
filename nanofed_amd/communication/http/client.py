"""aiohttp federation client (reference: nanofed/communication/http/client.py:24-242).

``fetch_global_model`` -> ``(state dict of tensors, round number)``;
``submit_update(model, metrics)`` -> server's ``accepted`` flag;
``check_server_status`` -> whether training is over; ``wait_for_completion``
polls it.  Every failure surfaces as ``NanoFedError``.  The ``timeout``
argument is applied to the session (the reference stores and ignores it).
Bodies are built and parsed with the native tensor/JSON codec.
"""

from __future__ import annotations

import asyncio
from dataclasses import dataclass
from typing import Any, TypeAlias

import aiohttp
import numpy as np
import torch

from nanofed_amd.communication import codec
from nanofed_amd.core import ModelProtocol, NanoFedError
from nanofed_amd.trainer.base import TrainingMetrics
from nanofed_amd.utils import Logger, get_current_time, log_exec

# what a wire body may hold for one parameter / for a whole model
TensorLike: TypeAlias = torch.Tensor | list[float] | list[list[float]] | float | int
ModelState: TypeAlias = dict[str, list[float] | list[list[float]]]


@dataclass(slots=True, frozen=True)
class ClientEndpoints:
    get_model: str = "/model"
    submit_update: str = "/update"
    get_status: str = "/status"


async def _body(response: Any) -> dict[str, Any]:
    """Parsed JSON body; raw bytes go through the native codec when the
    response object can provide them."""
    read = getattr(response, "read", None)
    if read is not None:
        return codec.loads_envelope(await read())
    return await response.json()


def _as_tensor(value: Any) -> torch.Tensor:
    return value if isinstance(value, torch.Tensor) else torch.tensor(value)


class HTTPClient:
    def __init__(
        self,
        server_url: str,
        client_id: str,
        endpoints: ClientEndpoints | None = None,
        timeout: int = 300,
    ) -> None:
        self._server_url = server_url.rstrip("/")
        self._client_id = client_id
        self._endpoints = endpoints or ClientEndpoints()
        self._logger = Logger()
        self._timeout = timeout
        self._current_round: int = 0
        self._session: aiohttp.ClientSession | None = None
        self._is_training_done: bool = False

    async def __aenter__(self) -> "HTTPClient":
        self._logger.info(f"Initializing HTTP client for {self._client_id}")
        if self._session is None:
            self._session = aiohttp.ClientSession(
                timeout=aiohttp.ClientTimeout(total=self._timeout)
            )
        return self

    async def __aexit__(self, exc_type, exc_val, exc_tb) -> None:
        self._logger.info(f"Closing HTTP client for {self._client_id}")
        if self._session:
            await self._session.close()
            self._session = None

    def _get_url(self, endpoint: str) -> str:
        return f"{self._server_url}{endpoint}"

    def _require_session(self) -> Any:
        if self._session is None:
            raise NanoFedError("Client session not initialized")
        return self._session

    @log_exec
    async def fetch_global_model(self) -> tuple[dict[str, torch.Tensor], int]:
        with self._logger.context("client.http"):
            session = self._require_session()
            try:
                url = self._get_url(self._endpoints.get_model)
                self._logger.info(f"Fetching global model from {url}...")
                async with session.get(url) as response:
                    if response.status != 200:
                        raise NanoFedError(
                            f"Server error while fetching model: {response.status}"
                        )
                    data = await _body(response)
                if data.get("status") != "success":
                    raise NanoFedError(
                        f"Error from server: {data.get('message', 'Unknown error')}"
                    )
                if "model_state" not in data or "round_number" not in data:
                    raise NanoFedError(
                        "Invalid server response: missing required fields"
                    )
                state = {
                    name: _as_tensor(value)
                    for name, value in data["model_state"].items()
                }
                self._current_round = data["round_number"]
                self._logger.info("Fetched global model.")
                return state, self._current_round
            except aiohttp.ClientError as exc:
                raise NanoFedError(f"HTTP error: {exc}") from exc
            except NanoFedError as exc:
                raise NanoFedError(f"Failed to fetch global model: {exc}") from exc
            except Exception as exc:
                raise NanoFedError(f"Failed to fetch global model: {exc}") from exc

    def _convert_tensor(self, value: Any) -> Any:
        if isinstance(value, torch.Tensor):
            return value.detach().cpu().tolist()
        if isinstance(value, (list, np.ndarray)):
            return value
        if isinstance(value, (int, float)):
            return [float(value)]
        return value

    @log_exec
    async def submit_update(
        self,
        model: ModelProtocol,
        metrics: dict[str, float] | TrainingMetrics,
        privacy_spent: Any = None,
    ) -> bool:
        with self._logger.context("client.http"):
            session = self._require_session()
            try:
                if self._is_training_done:
                    self._logger.info(
                        "Training is already complete. Skipped update."
                    )
                    return False
                if isinstance(metrics, TrainingMetrics):
                    metrics = metrics.to_dict()
                payload: dict[str, Any] = {
                    "client_id": self._client_id,
                    "round_number": self._current_round,
                    "model_state": dict(model.state_dict()),
                    "metrics": metrics,
                    "timestamp": get_current_time().isoformat(),
                }
                if privacy_spent is not None:
                    payload["privacy_spent"] = privacy_spent
                url = self._get_url(self._endpoints.submit_update)
                self._logger.info(
                    f"Submitting update to {url} for round {self._current_round}"
                )
                async with session.post(
                    url,
                    data=codec.dumps_envelope(payload),
                    headers={"Content-Type": "application/json"},
                ) as response:
                    if response.status != 200:
                        raise NanoFedError(f"Server error: {response.status}")
                    data = await _body(response)
                if data["status"] != "success":
                    raise NanoFedError(f"Error from server: {data['message']}")
                return bool(data["accepted"])
            except aiohttp.ClientError as exc:
                raise NanoFedError(f"HTTP error: {exc}") from exc
            except Exception as exc:
                raise NanoFedError(f"Failed to submit update: {exc}") from exc

    async def check_server_status(self) -> bool:
        session = self._require_session()
        try:
            url = self._get_url(self._endpoints.get_status)
            async with session.get(url) as response:
                if response.status != 200:
                    raise NanoFedError(
                        f"Failed to fetch server status: {response.status}"
                    )
                data = await _body(response)
            self._is_training_done = bool(data.get("is_training_done", False))
            return self._is_training_done
        except aiohttp.ClientError as exc:
            raise NanoFedError(f"HTTP error: {exc}") from exc

    async def wait_for_completion(self, poll_interval: float = 10) -> None:
        self._logger.info("Waiting for training to complete...")
        while not self._is_training_done:
            await self.check_server_status()
            if not self._is_training_done:
                await asyncio.sleep(poll_interval)
        self._logger.info("Training completed. Client can safely terminate.")
