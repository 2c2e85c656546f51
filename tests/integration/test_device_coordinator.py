"""``DeviceCoordinator``: the reference coordinator's calls and on-disk results
(nanofed/orchestration/coordinator.py:92-405) served by ``FederatedEngine``."""

import asyncio
import json
from pathlib import Path

import pytest
import torch

from nanofed_amd import CoordinatorConfig, DeviceCoordinator, TrainingConfig
from nanofed_amd.communication import InNodeClient, InNodeServer
from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.orchestration import RoundStatus, engine_config_from
from nanofed_amd.parallel import Communicator

ENGINE = dict(model="mnist_mlp", samples_per_client=64, eval_samples=32, dtype="fp32",
              backend="torch", model_kwargs={"hidden": 16})


def _config(tmp_path: Path, **kw) -> CoordinatorConfig:
    base = dict(num_rounds=3, min_clients=4, min_completion_rate=1.0, round_timeout=30,
                base_dir=tmp_path / "run")
    base.update(kw)
    return CoordinatorConfig(**base)


def test_engine_config_follows_the_reference_configs(tmp_path):
    cfg = engine_config_from(
        _config(tmp_path, min_completion_rate=0.5),
        TrainingConfig(epochs=3, batch_size=32, learning_rate=0.05), model="mnist_mlp",
    )
    assert (cfg.num_clients, cfg.num_rounds, cfg.local_epochs) == (4, 3, 3)
    assert (cfg.batch_size, cfg.learning_rate, cfg.min_completion_rate) == (32, 0.05, 0.5)
    assert cfg.round_timeout_s == 30.0 and cfg.required_clients() == 2


def test_rounds_metrics_versions_and_progress(tmp_path):
    config = _config(tmp_path)
    coordinator = DeviceCoordinator.create(
        config, TrainingConfig(epochs=1, batch_size=32, learning_rate=0.1),
        comm=Communicator(), **ENGINE,
    )
    seen, progress = [], []

    async def main():
        async for metrics in coordinator.start_training(progress.append):
            seen.append(metrics)

    asyncio.run(main())
    assert [m.round_id for m in seen] == [0, 1, 2]
    assert all(m.status is RoundStatus.COMPLETED and m.num_clients == 4 for m in seen)
    assert seen[-1].agg_metrics["loss"] < seen[0].agg_metrics["loss"]
    assert [p["current_round"] for p in progress] == [1, 2, 3]
    last = coordinator.training_progress
    assert last["status"] == "COMPLETED" and last["active_clients"] == 4
    assert last["total_rounds"] == 3
    assert last["global_metrics"]["loss"] == pytest.approx(
        sum(m.agg_metrics["loss"] for m in seen) / 3
    )
    base = config.base_dir
    assert coordinator.data_dir == base / "data" and coordinator.data_dir.is_dir()
    for r in range(3):
        record = json.loads((base / "metrics" / f"metrics_round_{r}.json").read_text())
        assert record["status"] == "COMPLETED" and record["num_clients"] == 4
        assert {c["client_id"] for c in record["client_metrics"]} == {f"client_{i}" for i in range(4)}
        assert sum(c["weight"] for c in record["client_metrics"]) == pytest.approx(1.0)
    # bootstrap version + one per round; the newest one is the engine's global model
    versions = coordinator.model_manager.list_versions()
    assert len(versions) == 4 and versions[-1].config["round_id"] == 2
    saved = torch.load(versions[-1].path, weights_only=True)
    live = coordinator.engine.global_state_dict()
    assert saved.keys() == live.keys()
    for k in saved:
        torch.testing.assert_close(saved[k], live[k], rtol=0, atol=0)
    coordinator.close()


def test_it_is_the_engine_that_runs(tmp_path):
    """Same seed, same config: the coordinator's model is ``FederatedEngine.run``'s."""
    config = _config(tmp_path, num_rounds=2)
    coordinator = DeviceCoordinator.create(
        config, TrainingConfig(epochs=1, batch_size=32, learning_rate=0.1),
        comm=Communicator(), **ENGINE,
    )

    async def main():
        return [m async for m in coordinator.start_training()]

    asyncio.run(main())
    plain = FederatedEngine(
        EngineConfig(num_clients=4, num_rounds=2, local_epochs=1, batch_size=32,
                     learning_rate=0.1, min_clients=4, round_timeout_s=30.0, **ENGINE),
        comm=Communicator(),
    )
    plain.run(2)
    torch.testing.assert_close(coordinator.engine.global_arena.data, plain.global_arena.data,
                               rtol=0, atol=0)


def test_a_round_below_quorum_raises_timeout_and_keeps_the_model(tmp_path):
    config = _config(tmp_path, num_rounds=1)
    engine = FederatedEngine(
        engine_config_from(config, TrainingConfig(epochs=1, batch_size=32, learning_rate=0.1), **ENGINE),
        comm=Communicator(), base_dir=config.base_dir,
        faults=FaultPlan(drop={0: {1}}),
    )
    coordinator = DeviceCoordinator(engine, config)
    before = engine.global_arena.data.clone()
    with pytest.raises(TimeoutError):
        asyncio.run(coordinator.train_round())
    assert coordinator.training_progress["status"] == "FAILED"
    torch.testing.assert_close(engine.global_arena.data, before, rtol=0, atol=0)
    # like the reference, a round that did not close leaves no metrics file and
    # the round counter where it was
    assert not (config.base_dir / "metrics" / "metrics_round_0.json").exists()
    assert coordinator.current_round == 0
    assert coordinator.round_metrics[-1].num_clients == 3


def test_a_failed_round_is_rolled_back_and_repeated(tmp_path):
    config = _config(tmp_path, num_rounds=3)
    engine = FederatedEngine(
        engine_config_from(config, TrainingConfig(epochs=1, batch_size=32, learning_rate=0.1), checkpoint_rounds=True,
                           **ENGINE),
        comm=Communicator(), base_dir=config.base_dir,
        faults=FaultPlan(fail={1: "HIP error: injected"}),
    )
    coordinator = DeviceCoordinator(engine, config)

    async def main():
        return [m.round_id async for m in coordinator.start_training()]

    assert asyncio.run(main()) == [0, 1, 2]
    assert engine.recoveries == 1


def test_readers_fetch_the_global_model_through_the_in_node_server(tmp_path):
    config = _config(tmp_path, num_rounds=2)
    server = InNodeServer("devcoord")
    coordinator = DeviceCoordinator.create(
        config, TrainingConfig(epochs=1, batch_size=32, learning_rate=0.1),
        server=server, comm=Communicator(), **ENGINE,
    )

    async def main():
        await server.start()
        fetched = []
        async with InNodeClient(server, "reader") as reader:
            async for _ in coordinator.start_training():
                state, round_number = await reader.fetch_global_model()
                fetched.append((round_number, {k: v.clone() for k, v in state.items()}))
            assert await reader.check_server_status() is True  # training is done
        await server.stop()
        return fetched

    fetched = asyncio.run(main())
    assert [r for r, _ in fetched] == [1, 2]
    live = coordinator.engine.global_state_dict()
    for k, v in fetched[-1][1].items():
        torch.testing.assert_close(v.cpu(), live[k], rtol=0, atol=0)
