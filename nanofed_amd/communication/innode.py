"""In-node transport: the ``HTTPServer`` / ``HTTPClient`` method surface
without sockets or JSON.

The reference's clients and coordinator are coroutines of one process that
nevertheless talk through a real TCP socket and encode every tensor as nested
Python float lists (nanofed/communication/http/client.py:103-211,
server.py:154-298): about 2 s per model fetch and 0.75 s per submission for a
4.8 MB model.  When clients and server live in the same process (or on GPUs
of the same node) none of that is needed:

* ``fetch_global_model`` hands out tensors that alias the server model's
  storage on its device (cloned on request), so a fetch is a pointer copy;
* ``submit_update`` stores the client's state dict as device tensors - an
  ``ArenaStateDict`` stays an arena, so the aggregator reduces it with one
  fused kernel;
* the protocol rules are unchanged: round check ("Invalid round number"),
  last-writer-wins per client, termination flag, status fields.

``InNodeServer`` plugs into ``Coordinator`` exactly like ``HTTPServer`` (it
exposes ``_updates``, ``set_coordinator``, ``advance_round``,
``stop_training``), so tutorial-style client loops only swap the two class
names.  The multi-GPU data plane (RCCL collectives over flat arenas) is
``nanofed_amd.parallel.Communicator`` + ``nanofed_amd.engine.FederatedEngine``.
"""

from __future__ import annotations

import asyncio
from typing import TYPE_CHECKING, Any

import torch

from nanofed_amd.core import ModelProtocol, NanoFedError, ServerProtocol
from nanofed_amd.trainer.base import TrainingMetrics
from nanofed_amd.utils import Logger, get_current_time, log_exec

if TYPE_CHECKING:
    from nanofed_amd.orchestration import Coordinator

_registry: dict[str, "InNodeServer"] = {}


class InNodeServer(ServerProtocol):
    def __init__(self, name: str = "node0", host: str = "in-node", port: int = 0) -> None:
        self._name = name
        self._host = host
        self._port = port
        self._logger = Logger()
        self._coordinator: "Coordinator | None" = None
        self._updates: dict[str, dict[str, Any]] = {}
        self._current_round: int = 0
        self._is_training_done = False
        self._lock = asyncio.Lock()
        self._started = False

    # -- ServerProtocol ---------------------------------------------------
    @property
    def host(self) -> str:
        return self._host

    @property
    def port(self) -> int:
        return self._port

    @property
    def url(self) -> str:
        return f"innode://{self._name}"

    def set_coordinator(self, coordinator: "Coordinator") -> None:
        self._coordinator = coordinator

    def advance_round(self, round_number: int) -> None:
        self._current_round = int(round_number)

    async def start(self) -> None:
        _registry[self.url] = self
        self._started = True

    async def stop(self) -> None:
        _registry.pop(self.url, None)
        self._started = False

    async def stop_training(self) -> None:
        self._is_training_done = True
        self._logger.info("Training completed. Clients will see the termination flag.")

    # -- what the HTTP handlers do, as plain calls -------------------------
    def get_model(self, clone: bool) -> dict[str, Any]:
        if self._coordinator is None:
            raise NanoFedError("Server error: Server not initialized with coordinator")
        if self._is_training_done:
            return {"status": "terminated", "message": "Training is complete",
                    "model_state": None, "round_number": -1}
        manager = self._coordinator.model_manager
        version = manager.current_version or manager.load_model()
        state = manager.model.state_dict()
        if clone:
            state = {k: v.detach().clone() for k, v in state.items()}
        return {"status": "success", "message": "Global model retrieved",
                "model_state": state, "round_number": self._current_round,
                "version_id": version.version_id}

    async def submit(self, update: dict[str, Any]) -> dict[str, Any]:
        async with self._lock:
            if update["round_number"] != self._current_round:
                self._logger.warning(
                    f"Update round mismatch: expected {self._current_round}, got "
                    f"{update['round_number']} from client {update['client_id']}"
                )
                return {"status": "error", "message": "Invalid round number", "accepted": False}
            cid = update["client_id"]
            self._updates[cid] = update
            return {"status": "success", "message": "Updated accepted", "accepted": True,
                    "update_id": f"update_{cid}_{self._current_round}"}

    def status(self) -> dict[str, Any]:
        return {"status": "success", "message": "Server is running",
                "current_round": self._current_round, "num_updates": len(self._updates),
                "is_training_done": self._is_training_done}


class InNodeClient:
    """Drop-in for ``HTTPClient`` against an ``InNodeServer``.

    ``server`` may be the server object or its ``url``.  ``copy_on_fetch``
    (default) gives the client private tensors, like a network fetch would;
    pass ``False`` to train directly on aliases of the global model's storage
    when the client promises to copy before mutating.
    """

    def __init__(
        self, server: "InNodeServer | str", client_id: str, copy_on_fetch: bool = True,
        timeout: int = 300,
    ) -> None:
        self._server_ref = server
        self._client_id = client_id
        self._copy = copy_on_fetch
        self._timeout = timeout
        self._logger = Logger()
        self._current_round = 0
        self._is_training_done = False
        self._open = False

    def _server(self) -> InNodeServer:
        if isinstance(self._server_ref, InNodeServer):
            return self._server_ref
        try:
            return _registry[self._server_ref]
        except KeyError as exc:
            raise NanoFedError(f"No in-node server is listening at {self._server_ref}") from exc

    async def __aenter__(self) -> "InNodeClient":
        self._open = True
        return self

    async def __aexit__(self, exc_type, exc_val, exc_tb) -> None:
        self._open = False

    def _require_open(self) -> None:
        if not self._open:
            raise NanoFedError("Client session not initialized")

    @log_exec
    async def fetch_global_model(self) -> tuple[dict[str, torch.Tensor], int]:
        self._require_open()
        with self._logger.context("client.innode"):
            data = self._server().get_model(self._copy)
            if data["status"] != "success":
                raise NanoFedError(f"Failed to fetch global model: {data['message']}")
            self._current_round = data["round_number"]
            return data["model_state"], self._current_round

    @log_exec
    async def submit_update(
        self,
        model: ModelProtocol | dict[str, torch.Tensor],
        metrics: dict[str, float] | TrainingMetrics,
        privacy_spent: Any = None,
    ) -> bool:
        self._require_open()
        if self._is_training_done:
            return False
        if isinstance(metrics, TrainingMetrics):
            metrics = metrics.to_dict()
        state = model if isinstance(model, dict) else model.state_dict()
        if not hasattr(state, "flat"):
            state = {k: v.detach() for k, v in state.items()}
        reply = await self._server().submit(
            {"client_id": self._client_id, "round_number": self._current_round,
             "model_state": state, "metrics": dict(metrics),
             "timestamp": get_current_time().isoformat(), "privacy_spent": privacy_spent}
        )
        if reply["status"] != "success":
            raise NanoFedError(f"Failed to submit update: {reply['message']}")
        return bool(reply["accepted"])

    async def check_server_status(self) -> bool:
        self._require_open()
        self._is_training_done = bool(self._server().status()["is_training_done"])
        return self._is_training_done

    async def wait_for_completion(self, poll_interval: float = 10) -> None:
        while not await self.check_server_status():
            await asyncio.sleep(poll_interval)
