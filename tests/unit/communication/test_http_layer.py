"""HTTP client against a fake session, HTTP server handlers through aiohttp's
test client, and the tensor/JSON codec (nanofed/communication/http/*)."""

import asyncio
import json
from unittest.mock import MagicMock

import pytest
import torch
import torch.nn as nn
from aiohttp.test_utils import TestClient, TestServer

from nanofed_amd.communication import HTTPClient, HTTPServer, codec
from nanofed_amd.communication.http import ClientEndpoints, ServerEndpoints
from nanofed_amd.core import NanoFedError
from nanofed_amd.trainer import TrainingMetrics


class FakeResponse:
    def __init__(self, status, payload):
        self.status, self._payload = status, payload

    async def json(self):
        return self._payload

    async def __aenter__(self):
        return self

    async def __aexit__(self, *exc):
        return False


class FakeSession:
    def __init__(self, routes):
        self.routes, self.posted = routes, []

    def get(self, url):
        return self.routes[url.rsplit("/", 1)[-1]]

    def post(self, url, data=None, headers=None, json=None):
        self.posted.append(json if json is not None else __import__("json").loads(data))
        return self.routes[url.rsplit("/", 1)[-1]]

    async def close(self):
        pass


def run(coro):
    return asyncio.run(coro)


def test_endpoints_default_paths():
    assert (ClientEndpoints().get_model, ClientEndpoints().submit_update,
            ClientEndpoints().get_status) == ("/model", "/update", "/status")
    assert ServerEndpoints().get_model == "/model"


def test_client_fetch_parses_state_and_round():
    async def go():
        client = HTTPClient("http://h:1/", "c1")
        client._session = FakeSession({"model": FakeResponse(200, {
            "status": "success", "message": "ok", "timestamp": "t",
            "model_state": {"fc.weight": [[1.0, 2.0]], "fc.bias": [0.5]},
            "round_number": 3, "version_id": "v"})})
        state, rnd = await client.fetch_global_model()
        assert rnd == 3 and client._current_round == 3
        assert torch.equal(state["fc.weight"], torch.tensor([[1.0, 2.0]]))
        assert state["fc.bias"].dtype == torch.float32
    run(go())


@pytest.mark.parametrize("resp, text", [
    (FakeResponse(500, {}), "Server error"),
    (FakeResponse(200, {"status": "terminated", "message": "Training is complete",
                        "model_state": None, "round_number": -1}), "Training is complete"),
    (FakeResponse(200, {"status": "success"}), "missing required fields"),
])
def test_client_fetch_failures_raise_nanofed_error(resp, text):
    async def go():
        client = HTTPClient("http://h:1", "c1")
        client._session = FakeSession({"model": resp})
        with pytest.raises(NanoFedError, match=text):
            await client.fetch_global_model()
    run(go())


def test_client_submit_sends_the_wire_schema():
    async def go():
        client = HTTPClient("http://h:1", "c7")
        session = FakeSession({"update": FakeResponse(200, {
            "status": "success", "message": "Updated accepted", "timestamp": "t",
            "update_id": "update_c7_0", "accepted": True})})
        client._session = session
        model = nn.Linear(2, 1)
        ok = await client.submit_update(model, TrainingMetrics(0.4, 0.9, 1, 5, 64))
        assert ok is True
        body = session.posted[0]
        assert set(body) == {"client_id", "round_number", "model_state", "metrics", "timestamp"}
        assert body["client_id"] == "c7" and body["round_number"] == 0
        assert body["metrics"] == {"loss": 0.4, "accuracy": 0.9, "samples_processed": 64}
        # nested lists of numbers that read back to the same fp32 values
        assert torch.equal(torch.tensor(body["model_state"]["weight"]), model.weight.detach())
        session.routes["update"] = FakeResponse(500, {})
        with pytest.raises(NanoFedError, match="Server error"):
            await client.submit_update(model, {"loss": 1.0})
        client._is_training_done = True
        assert await client.submit_update(model, {"loss": 1.0}) is False
    run(go())


def test_client_status_and_wait_for_completion():
    async def go():
        client = HTTPClient("http://h:1", "c1")
        client._session = FakeSession({"status": FakeResponse(200, {"is_training_done": False})})
        assert await client.check_server_status() is False
        client._session.routes["status"] = FakeResponse(200, {"is_training_done": True})
        await client.wait_for_completion(poll_interval=0.01)
        assert client._is_training_done
        with pytest.raises(NanoFedError, match="not initialized"):
            await HTTPClient("http://h:1", "c2").fetch_global_model()
    run(go())


# -- server ------------------------------------------------------------------
class ListModel:
    def state_dict(self):
        return {"fc.weight": [[1.0, 2.0], [3.0, 4.0]], "fc.bias": [0.5, 0.6]}


def mock_coordinator():
    coord = MagicMock()
    coord.model_manager.current_version.version_id = "model_v_test"
    coord.model_manager.model = ListModel()
    return coord


def with_server(fn, coordinator=True):
    async def go():
        server = HTTPServer("127.0.0.1", 0)
        if coordinator:
            server.set_coordinator(mock_coordinator())
        async with TestClient(TestServer(server._app)) as tc:
            await fn(server, tc)
    run(go())


def test_server_properties_and_liveness():
    server = HTTPServer("0.0.0.0", 8123, max_request_size=1234)
    assert (server.host, server.port, server.url) == ("0.0.0.0", 8123, "http://0.0.0.0:8123")

    async def check(server, tc):
        resp = await tc.get("/test")
        assert resp.status == 200 and await resp.text() == "Server is running"
    with_server(check)


def test_get_model_payload_and_termination():
    async def check(server, tc):
        resp = await tc.get("/model")
        data = await resp.json()
        assert resp.status == 200 and data["status"] == "success"
        assert data["message"] == "Global model retrieved"
        assert data["model_state"] == ListModel().state_dict()
        assert data["round_number"] == 0 and data["version_id"] == "model_v_test"
        await server.stop_training()
        data = await (await tc.get("/model")).json()
        assert data["status"] == "terminated" and data["model_state"] is None
        assert data["round_number"] == -1 and data["message"] == "Training is complete"
    with_server(check)


def test_get_model_without_coordinator_is_a_500():
    async def check(server, tc):
        resp = await tc.get("/model")
        data = await resp.json()
        assert resp.status == 500 and data["status"] == "error"
        assert data["message"] == "Server not initialized with coordinator"
    with_server(check, coordinator=False)


def test_submit_update_rules_and_status():
    body = {"client_id": "c1", "round_number": 0, "model_state": {"w": [1.0, 2.0]},
            "metrics": {"loss": 0.1}, "timestamp": "2024-01-01T00:00:00+00:00"}

    async def check(server, tc):
        resp = await tc.post("/update", json=body)
        data = await resp.json()
        assert resp.status == 200 and data["accepted"] is True
        assert data["update_id"] == "update_c1_0" and data["message"] == "Updated accepted"
        assert "c1" in server._updates and "privacy_spent" in server._updates["c1"]
        # last writer wins per client
        await tc.post("/update", json={**body, "metrics": {"loss": 0.05}})
        assert len(server._updates) == 1 and server._updates["c1"]["metrics"]["loss"] == 0.05
        partial = {k: v for k, v in body.items() if k not in ("metrics", "timestamp")}
        resp = await tc.post("/update", json=partial)
        assert resp.status == 400 and "Missing keys: metrics, timestamp" in (await resp.json())["message"]
        resp = await tc.post("/update", json={**body, "round_number": 5})
        assert resp.status == 400 and (await resp.json())["message"] == "Invalid round number"
        # stale updates are refused once the coordinator moved on
        server.advance_round(1)
        resp = await tc.post("/update", json=body)
        assert resp.status == 400
        st = await (await tc.get("/status")).json()
        assert st["current_round"] == 1 and st["num_updates"] == 1
        assert st["is_training_done"] is False and st["message"] == "Server is running"
    with_server(check)


# -- codec ---------------------------------------------------------------------
def test_codec_round_trip_matches_plain_json():
    torch.manual_seed(0)
    state = {"a": torch.randn(3, 4), "b": torch.randn(5), "s": torch.tensor(2.5),
             "i": torch.arange(6).reshape(2, 3)}
    body = codec.dumps_envelope({"client_id": "x", "round_number": 1, "model_state": state,
                                 "metrics": {"loss": 0.25}})
    plain = json.loads(body)
    assert plain["client_id"] == "x" and plain["metrics"] == {"loss": 0.25}
    assert torch.equal(torch.tensor(plain["model_state"]["a"]), state["a"])
    assert plain["model_state"]["i"] == [[0, 1, 2], [3, 4, 5]]
    back = codec.loads_envelope(body)
    for k, v in state.items():
        got = torch.as_tensor(back["model_state"][k])
        assert torch.equal(got.to(v.dtype).reshape(v.shape), v)
    # bodies produced by any JSON writer are accepted
    other = json.dumps({"model_state": {"w": [[1, 2.5], [3e-3, -4]]}, "round_number": 0}).encode()
    got = codec.loads_envelope(other)
    assert torch.equal(torch.as_tensor(got["model_state"]["w"], dtype=torch.float32),
                       torch.tensor([[1, 2.5], [3e-3, -4]]))


def test_server_encodes_the_global_model_once_until_it_changes():
    """Every client of a round fetches the same model: the JSON text is kept and only
    rebuilt when a tensor of the model is replaced or written in place."""
    import torch
    import torch.nn as nn

    from nanofed_amd.communication import codec
    from nanofed_amd.communication.http.server import HTTPServer

    server = HTTPServer("127.0.0.1", 0)
    model = nn.Linear(4, 2)
    first = server._encoded_model(model)
    again = server._encoded_model(model)
    assert isinstance(first, codec.EncodedState) and again.text is first.text
    body = codec.dumps_envelope({"status": "success", "model_state": first, "round_number": 3})
    back = codec.loads_envelope(body)
    assert back["round_number"] == 3
    assert torch.equal(back["model_state"]["weight"], model.weight.detach())
    with torch.no_grad():
        model.weight.add_(1.0)  # in place: same storage, new version counter
    changed = server._encoded_model(model)
    assert changed.text is not first.text
    back = codec.loads_envelope(codec.dumps_envelope({"model_state": changed}))
    assert torch.equal(back["model_state"]["weight"], model.weight.detach())
    model.load_state_dict({k: v * 0 for k, v in model.state_dict().items()})
    zero = server._encoded_model(model)
    assert zero.text is not changed.text
    back = codec.loads_envelope(codec.dumps_envelope({"model_state": zero}))
    assert float(back["model_state"]["weight"].abs().sum()) == 0.0
