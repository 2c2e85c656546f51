"""Mixed rounds: clients that train outside the engine take part in a device round as extra
arena rows (reference: a round is made of the updates that were POSTed,
nanofed/orchestration/coordinator.py:205-245,307-322)."""

import asyncio
from dataclasses import replace

import pytest
import torch

from nanofed_amd.communication.innode import InNodeClient, InNodeServer
from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.engine.data import build_federated_data
from nanofed_amd.orchestration import CoordinatorConfig
from nanofed_amd.orchestration.device import DeviceCoordinator
from nanofed_amd.parallel import Communicator

SIZES = [96, 64, 64, 32, 48, 80]


def cfg(**kw):
    base = dict(model="mnist_mlp", num_clients=6, samples_per_client=SIZES, batch_size=32,
                local_epochs=1, eval_samples=64, dtype="fp32", backend="torch",
                model_kwargs={"hidden": 16}, learning_rate=0.05)
    base.update(kw)
    return EngineConfig(**base)


def _four_of_six():
    """The data of the six-client run, and the same with the first four shards only."""
    data = build_federated_data(cfg())
    return data, replace(data, shards=data.shards[:4])


def test_engine_with_two_external_rows_equals_the_six_client_engine():
    data, first_four = _four_of_six()
    ref = FederatedEngine(cfg(), comm=Communicator(), data=data)
    ref_rep = ref.run_round()
    mixed = FederatedEngine(cfg(num_clients=4, samples_per_client=SIZES[:4], external_slots=2),
                            comm=Communicator(), data=first_four)
    # the outside clients hand in what clients 4 and 5 of the reference run trained
    for slot, k in enumerate((4, 5)):
        mixed.submit_external(slot, ref.pool[k].state_dict(), SIZES[k],
                              {"loss": ref_rep.client_metrics[k]["metrics"]["loss"],
                               "accuracy": ref_rep.client_metrics[k]["metrics"]["accuracy"]},
                              client_id=f"outside_{k}")
    rep = mixed.run_round()
    assert rep.status == "COMPLETED" and rep.num_clients == 6
    assert torch.equal(mixed.global_arena.data, ref.global_arena.data)
    assert [c["client_id"] for c in rep.client_metrics[4:]] == ["outside_4", "outside_5"]
    assert [c["weight"] for c in rep.client_metrics] == [c["weight"] for c in ref_rep.client_metrics]
    assert rep.agg_metrics["loss"] == pytest.approx(ref_rep.agg_metrics["loss"], rel=1e-6)
    # next round: nobody submitted - the slots take no part, four clients make the round
    rep2 = mixed.run_round()
    assert rep2.num_clients == 4 and [c["valid"] for c in rep2.client_metrics] == [True] * 4 + [False] * 2
    assert sum(c["weight"] for c in rep2.client_metrics[:4]) == pytest.approx(1.0)


def test_external_rows_pass_through_the_same_screening():
    _, first_four = _four_of_six()
    eng = FederatedEngine(cfg(num_clients=4, samples_per_client=SIZES[:4], external_slots=2,
                              max_update_norm=50.0),
                          comm=Communicator(), data=first_four)
    good = eng.global_arena.data.clone() + 0.01
    bad = eng.global_arena.data.clone()
    bad[3] = float("nan")
    eng.submit_external(0, good, 64)
    eng.submit_external(1, bad, 64)
    rep = eng.run_round()
    assert [c["valid"] for c in rep.client_metrics[4:]] == [True, False]
    assert torch.isfinite(eng.global_arena.data).all()
    # far away from the global model: rejected by the norm bound
    eng.submit_external(0, eng.global_arena.data.clone() + 10.0, 64)
    rep = eng.run_round()
    assert not rep.client_metrics[4]["valid"] and rep.num_clients == 4
    with pytest.raises(Exception, match="out of range"):
        eng.submit_external(2, good, 1)
    with pytest.raises(Exception, match="secure_agg"):
        FederatedEngine(cfg(external_slots=1, secure_agg="pairwise"), comm=Communicator())


def test_device_coordinator_round_with_in_node_submitters(tmp_path):
    """Four resident clients + two InNodeClient submitters through the coordinator's
    server: the round equals the six-client engine round."""
    data, first_four = _four_of_six()
    ref = FederatedEngine(cfg(), comm=Communicator(), data=data)
    ref.run_round()

    async def main():
        server = InNodeServer("mixed")
        await server.start()
        coord = DeviceCoordinator.create(
            CoordinatorConfig(num_rounds=1, min_clients=4, min_completion_rate=1.0, round_timeout=30,
                              base_dir=tmp_path),
            server=server, data=first_four, external_updates=2,
            **{k: v for k, v in vars(cfg(num_clients=4, samples_per_client=SIZES[:4],
                                         external_slots=2)).items()
               if k not in ("num_clients", "num_rounds", "min_clients", "min_completion_rate",
                            "round_timeout_s")})

        async def outsider(k):
            async with InNodeClient(server, f"outside_{k}") as client:
                state, rnd = await client.fetch_global_model()
                assert rnd == 0
                # (its training: the reference run's client k started from the same model)
                ok = await client.submit_update(ref.pool[k].state_dict(),
                                                {"loss": 0.5, "accuracy": 0.5, "num_samples": SIZES[k]})
                assert ok

        async def rounds():
            return [m async for m in coord.start_training()]

        _, _, metrics = await asyncio.gather(outsider(4), outsider(5), rounds())
        await server.stop()
        return coord, metrics

    coord, metrics = asyncio.run(main())
    assert metrics[0].num_clients == 6
    assert torch.equal(coord.engine.global_arena.data, ref.global_arena.data)
    assert {"outside_4", "outside_5"} <= set(coord._clients)
    coord.close()


def test_the_event_loop_keeps_running_while_a_round_trains(tmp_path):
    """With a server attached the round runs on the coordinator's worker thread: a
    coroutine that polls the server's status gets its turns during the round."""
    async def main():
        server = InNodeServer("busy")
        await server.start()
        coord = DeviceCoordinator.create(
            CoordinatorConfig(num_rounds=1, min_clients=4, min_completion_rate=1.0, round_timeout=30,
                              base_dir=tmp_path),
            server=server, model="mnist_mlp", dataset="synthetic_mnist", samples_per_client=2048,
            batch_size=32, local_epochs=2, eval_samples=64, dtype="fp32", backend="torch")
        assert coord._background
        seen = []
        done = asyncio.Event()

        async def poller():
            while not done.is_set():
                seen.append(server.status()["current_round"])
                await asyncio.sleep(0.002)

        async def one_round():
            try:
                return await coord.train_round()
            finally:
                done.set()

        _, metrics = await asyncio.gather(poller(), one_round())
        await server.stop()
        coord.close()
        return seen, metrics

    seen, metrics = asyncio.run(main())
    assert metrics.status.name == "COMPLETED"
    assert len(seen) >= 5, "the poller never ran during the round"


MIXED_WORKER = r"""
import sys, torch
sys.path.insert(0, {root!r})
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
comm = Communicator.from_env(backend="gloo", device="cpu")
eng = FederatedEngine(EngineConfig(**{cfg!r}), comm=comm)
reports = []
for r in range(2):
    if comm.is_root:
        for slot in range(2 - r):  # two outside updates in round 0, one in round 1
            gen = torch.Generator().manual_seed(100 * r + slot)
            delta = 0.01 * torch.randn(eng.layout.numel, generator=gen)
            eng.submit_external(slot, eng.global_arena.data + delta, 40 + 8 * slot)
    reports.append(eng.run_round())
if comm.is_root:
    torch.save({{"flat": eng.global_arena.data, "n": [r.num_clients for r in reports],
                "weights": [c["weight"] for c in reports[-1].client_metrics]}}, {out!r})
eng.close()
comm.finish(0)
"""


@pytest.mark.parametrize("nproc", [2, 3])
def test_mixed_rounds_on_several_ranks_equal_the_one_process_run(tmp_path, nproc):
    """Rank 0 takes the outside updates and broadcasts them; every rank adds its slice of
    them after the resident clients' slices, in one order: the model has the bits of the
    one-process run (six resident clients on 2 and on 3 ranks)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(__file__))
    from test_multiprocess_engine import _env, _free_port

    conf = dict(vars(cfg(external_slots=2)))

    def run(n, name):
        out = tmp_path / f"{name}.pt"
        script = tmp_path / f"{name}.py"
        script.write_text(MIXED_WORKER.format(root=root, cfg=conf, out=str(out)))
        port = _free_port()
        procs = [subprocess.Popen([sys.executable, str(script)], cwd=root, stderr=subprocess.PIPE,
                                  text=True, env=_env(RANK=str(k), WORLD_SIZE=str(n),
                                                      LOCAL_RANK=str(k), MASTER_PORT=str(port)))
                 for k in range(n)]
        for p in procs:
            _, err = p.communicate(timeout=300)
            assert p.returncode == 0, err[-2000:]
        return torch.load(out, weights_only=False)

    one, many = run(1, "one"), run(nproc, "many")
    assert one["n"] == many["n"] == [8, 7]
    assert many["weights"] == one["weights"]
    assert torch.equal(many["flat"], one["flat"])
