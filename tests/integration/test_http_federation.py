"""Real aiohttp server + clients on localhost, driven by the Coordinator.

Covers the plumbing configuration "MNIST 2-layer MLP, 2 CPU clients, 1 FedAvg
round over localhost HTTP", stale-round rejection, the quorum rule and the
on-disk artefacts (model versions, round metrics)."""

import asyncio
import json
import socket

import pytest
import torch

from nanofed_amd import (
    Coordinator,
    CoordinatorConfig,
    FedAvgAggregator,
    HTTPClient,
    HTTPServer,
    ModelManager,
    TorchTrainer,
    TrainingConfig,
    coordinate,
)
from nanofed_amd.core import NanoFedError
from nanofed_amd.data import synthetic_mnist
from nanofed_amd.data.mnist import normalize_mnist
from nanofed_amd.models import MNISTMLP
from nanofed_amd.orchestration import RoundStatus


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def shard(n, seed):
    x, y = synthetic_mnist(n, seed=seed)
    ds = torch.utils.data.TensorDataset(normalize_mnist(x), y)
    return torch.utils.data.DataLoader(ds, batch_size=32, shuffle=True)


async def run_client(cid, url, n, seed, rounds_seen):
    trainer = TorchTrainer(TrainingConfig(epochs=1, batch_size=32, learning_rate=0.1))
    data = shard(n, seed)
    async with HTTPClient(url, cid) as client:
        while True:
            if await client.check_server_status():
                break
            try:
                state, rnd = await client.fetch_global_model()
            except NanoFedError:
                break
            if rnd in rounds_seen[cid]:
                await asyncio.sleep(0.02)
                continue
            model = MNISTMLP(hidden=32)
            model.load_state_dict(state)
            opt = torch.optim.SGD(model.parameters(), lr=0.1)
            metrics = trainer.train_epoch(model, data, opt, 0)
            if await client.submit_update(model, metrics):
                rounds_seen[cid].add(rnd)


@pytest.mark.parametrize("rounds", [1, 2])
def test_two_cpu_clients_over_localhost_http(tmp_path, rounds):
    async def go():
        port = free_port()
        server = HTTPServer("127.0.0.1", port)
        await server.start()
        try:
            torch.manual_seed(0)
            manager = ModelManager(MNISTMLP(hidden=32))
            coord = Coordinator(
                manager, FedAvgAggregator(), server,
                CoordinatorConfig(num_rounds=rounds, min_clients=2, min_completion_rate=1.0,
                                  round_timeout=60, base_dir=tmp_path),
            )
            initial = {k: v.clone() for k, v in manager.model.state_dict().items()}
            seen = {"client_1": set(), "client_2": set()}
            await asyncio.gather(
                coordinate(coord),
                run_client("client_1", server.url, 192, 1, seen),
                run_client("client_2", server.url, 96, 2, seen),
            )
            return coord, manager, initial, seen
        finally:
            await server.stop()

    coord, manager, initial, seen = asyncio.run(go())
    assert coord.current_round == rounds
    assert coord.training_progress["status"] == RoundStatus.COMPLETED.name
    assert coord.training_progress["active_clients"] == 2
    assert seen["client_1"] == set(range(rounds)) == seen["client_2"]
    assert any(not torch.equal(initial[k], v) for k, v in manager.model.state_dict().items())
    # artefacts: bootstrap + one version per round, one metrics file per round
    versions = sorted((tmp_path / "models" / "configs").glob("model_v_*.json"))
    assert len(versions) == rounds + 1
    assert len(list((tmp_path / "models" / "models").glob("model_v_*.pt"))) == rounds + 1
    for r in range(rounds):
        rec = json.loads((tmp_path / "metrics" / f"metrics_round_{r}.json").read_text())
        assert rec["round_id"] == r and rec["num_clients"] == 2 and rec["status"] == "COMPLETED"
        weights = {c["client_id"]: c["weight"] for c in rec["client_metrics"]}
        assert weights["client_1"] == pytest.approx(2 / 3) and weights["client_2"] == pytest.approx(1 / 3)
        assert set(rec["agg_metrics"]) >= {"loss", "accuracy", "samples_processed"}
    last = json.loads(versions[-1].read_text())
    assert last["config"]["round_id"] == rounds - 1 and last["config"]["num_clients"] == 2
    state = torch.load(tmp_path / "models" / "models" / f"{last['version_id']}.pt", weights_only=True)
    assert set(state) == {"fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"}


def test_round_times_out_without_clients_and_quorum_is_never_zero(tmp_path):
    async def go():
        server = HTTPServer("127.0.0.1", free_port())
        coord = Coordinator(
            ModelManager(MNISTMLP(hidden=8)), FedAvgAggregator(), server,
            CoordinatorConfig(num_rounds=1, min_clients=1, min_completion_rate=0.5,
                              round_timeout=0.2, base_dir=tmp_path),
        )
        assert coord._required_clients() == 1
        with pytest.raises(TimeoutError):
            await coord.train_round()
        assert coord.training_progress["status"] == "FAILED"
    asyncio.run(go())


def test_client_without_coordinator_gets_a_clean_error():
    async def go():
        server = HTTPServer("127.0.0.1", free_port())
        await server.start()
        try:
            async with HTTPClient(server.url, "c") as client:
                with pytest.raises(NanoFedError):
                    await client.fetch_global_model()
        finally:
            await server.stop()
    asyncio.run(go())
