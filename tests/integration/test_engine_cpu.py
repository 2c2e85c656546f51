"""The device-plane engine on CPU (torch trainer): rounds, quorum, fault
injection, persistence formats, and a 2-process gloo run that must equal the
single-process result."""

import json
import os
import subprocess
import sys

import pytest
import torch

from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine, clients_of_rank
from nanofed_amd.parallel import Communicator

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cfg(**kw):
    base = dict(model="mnist_mlp", num_clients=4, samples_per_client=96, batch_size=32,
                local_epochs=1, eval_samples=200, dtype="fp32", backend="torch",
                model_kwargs={"hidden": 16}, learning_rate=0.1)
    base.update(kw)
    return EngineConfig(**base)


def test_client_placement():
    assert clients_of_rank(8, 1, 0) == list(range(8))
    assert [clients_of_rank(8, 4, r) for r in range(4)] == [[0, 1], [2, 3], [4, 5], [6, 7]]
    assert sum(len(clients_of_rank(7, 3, r)) for r in range(3)) == 7


def test_rounds_learn_and_fedavg_equals_the_weighted_mean_of_client_models():
    eng = FederatedEngine(cfg(samples_per_client=[160, 96, 64, 32]), comm=Communicator())
    rep0 = eng.run_round()
    assert rep0.status == "COMPLETED" and rep0.num_clients == 4
    w = [c["weight"] for c in rep0.client_metrics]
    assert w == pytest.approx([160 / 352, 96 / 352, 64 / 352, 32 / 352])
    want = sum(wk * row for wk, row in zip(w, eng.pool.rows()))
    torch.testing.assert_close(eng.global_arena.data, want, rtol=1e-5, atol=1e-6)
    for _ in range(5):
        rep = eng.run_round()
    assert rep.agg_metrics["loss"] < rep0.agg_metrics["loss"]
    assert eng.evaluate()["accuracy"] > 0.2  # chance is 0.1; 352 samples, 6 short rounds
    assert eng.round_id == 6


def test_dropped_and_poisoned_clients_are_masked_out():
    faults = FaultPlan(drop={0: {3}}, poison={1: {0}})
    eng = FederatedEngine(cfg(), comm=Communicator(), faults=faults)
    r0, r1 = eng.run_round(), eng.run_round()
    assert r0.num_clients == 3 and not r0.client_metrics[3]["valid"]
    assert r0.client_metrics[3]["weight"] == 0.0
    assert r1.num_clients == 3 and not r1.client_metrics[0]["valid"]
    assert torch.isfinite(eng.global_arena.data).all()


def test_round_fails_below_quorum_and_keeps_the_old_model():
    faults = FaultPlan(drop={0: {1, 2, 3}})
    eng = FederatedEngine(cfg(min_clients=4, min_completion_rate=0.5), comm=Communicator(),
                          faults=faults)
    before = eng.global_arena.data.clone()
    rep = eng.run_round()
    assert rep.status == "FAILED" and eng.round_id == 0
    assert torch.equal(eng.global_arena.data, before)
    assert eng.run_round().status == "FAILED"  # the plan still names round 0
    eng.faults = FaultPlan()
    assert eng.run_round().status == "COMPLETED" and eng.round_id == 1


def test_update_norm_bound_rejects_outliers():
    eng = FederatedEngine(cfg(max_update_norm=1e-6), comm=Communicator())
    assert eng.run_round().status == "FAILED"


def test_persistence_uses_the_reference_formats(tmp_path):
    eng = FederatedEngine(cfg(), comm=Communicator(), base_dir=tmp_path)
    eng.run(2)
    configs = sorted((tmp_path / "models" / "configs").glob("*.json"))
    assert len(configs) == 3  # bootstrap + 2 rounds
    rec = json.loads(configs[-1].read_text())
    assert rec["config"]["round_id"] == 1 and rec["config"]["status"] == "AGGREGATING"
    state = torch.load(tmp_path / "models" / "models" / f"{rec['version_id']}.pt", weights_only=True)
    want = eng.global_state_dict()
    assert all(torch.equal(state[k], v) for k, v in want.items())
    m = json.loads((tmp_path / "metrics" / "metrics_round_1.json").read_text())
    assert m["status"] == "COMPLETED" and len(m["client_metrics"]) == 4


def test_dp_batch_clip_path_runs_and_dirichlet_sharding():
    from nanofed_amd.engine import DPConfig

    eng = FederatedEngine(
        cfg(partition="dirichlet", dirichlet_alpha=0.3, samples_per_client=128,
            dp=DPConfig(max_gradient_norm=1.0, noise_multiplier=0.5, mode="batch")),
        comm=Communicator(),
    )
    rep = eng.run_round()
    assert rep.status == "COMPLETED"
    assert sum(c["metrics"]["samples_processed"] for c in rep.client_metrics) == 4 * 128


def _free_port() -> int:
    """A TCP port that is free right now (a fixed one can still be in TIME_WAIT from
    the previous parametrisation)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = r"""
import sys, json, torch
sys.path.insert(0, {root!r})
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
comm = Communicator.from_env(backend="gloo", device="cpu")
cfg = EngineConfig(model="mnist_mlp", num_clients=4, samples_per_client=[96, 64, 64, 32],
                   batch_size=32, local_epochs=1, eval_samples=64, dtype="fp32", backend="torch",
                   model_kwargs={{"hidden": 16}}, learning_rate=0.1, reduce_algo={algo!r})
eng = FederatedEngine(cfg, comm=comm)
reps = eng.run(2)
if comm.is_root:
    torch.save({{"flat": eng.global_arena.data, "loss": reps[-1].agg_metrics["loss"],
                "n": reps[-1].num_clients}}, {out!r})
comm.finish(0)
"""


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_two_process_gloo_run_matches_single_process(tmp_path, algo):
    out = tmp_path / "dist.pt"
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(out), algo=algo))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    proc = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
        capture_output=True, text=True, timeout=300, env=env,
    )
    assert proc.returncode == 0, proc.stderr[-2000:]
    dist = torch.load(out, weights_only=True)
    single = FederatedEngine(
        EngineConfig(model="mnist_mlp", num_clients=4, samples_per_client=[96, 64, 64, 32],
                     batch_size=32, local_epochs=1, eval_samples=64, dtype="fp32",
                     backend="torch", model_kwargs={"hidden": 16}, learning_rate=0.1),
        comm=Communicator(),
    )
    reps = single.run(2)
    assert dist["n"] == 4
    torch.testing.assert_close(dist["flat"], single.global_arena.data, rtol=1e-5, atol=1e-6)
    assert dist["loss"] == pytest.approx(reps[-1].agg_metrics["loss"], rel=1e-4)


# -- the next round queued before the host has finished the current one -------------------
_PREFETCH_CFG = dict(model="mnist_mlp", num_clients=4, samples_per_client=[96, 64, 64, 32],
                     batch_size=32, local_epochs=2, eval_samples=64, dtype="fp32",
                     backend="torch", model_kwargs={"hidden": 16}, learning_rate=0.1)


def _run_prefetch(rounds=4, **kw):
    eng = FederatedEngine(EngineConfig(**_PREFETCH_CFG, **kw), comm=Communicator())
    reports = eng.run(rounds)
    return eng, reports


def test_queueing_the_next_round_early_changes_nothing_but_the_order_of_host_work():
    early, ra = _run_prefetch(prefetch_rounds=True)
    plain, rb = _run_prefetch(prefetch_rounds=False)
    assert torch.equal(early.global_arena.data, plain.global_arena.data)
    assert [r.round_id for r in ra] == [0, 1, 2, 3] and early.round_id == 4
    assert [r.agg_metrics for r in ra] == [r.agg_metrics for r in rb]
    # every round reports its own phases, the queued ones included
    for rep in ra:
        assert {"broadcast", "local_train", "reduce"} <= set(rep.phases_ms)
    assert early._prelaunched is None  # the last round of run() queues nothing


def test_a_round_queued_for_nothing_is_trained_again_from_the_unchanged_model():
    # every client is screened out, so every round loses its quorum and the round counter
    # never advances: the round queued early (r + 1) never comes
    early, ra = _run_prefetch(rounds=3, prefetch_rounds=True, max_update_norm=1e-9)
    plain, rb = _run_prefetch(rounds=3, prefetch_rounds=False, max_update_norm=1e-9)
    assert [r.status for r in ra] == ["FAILED"] * 3 and early.round_id == 0
    assert torch.equal(early.global_arena.data, plain.global_arena.data)
    assert [r.client_metrics for r in ra] == [r.client_metrics for r in rb]


def test_single_rounds_queue_nothing_unless_asked_and_a_resume_drops_what_was_queued(tmp_path):
    eng = FederatedEngine(EngineConfig(**_PREFETCH_CFG), comm=Communicator(), base_dir=tmp_path)
    eng.run_round()
    assert eng._prelaunched is None
    eng.run_round(prefetch_next=True)
    assert eng._prelaunched is not None and eng._prelaunched[0] == 2
    assert eng.resume() == 2 and eng._prelaunched is None
    after = eng.run_round()
    ref = FederatedEngine(EngineConfig(**_PREFETCH_CFG), comm=Communicator())
    ref.run(3)
    assert after.round_id == 2 and torch.equal(eng.global_arena.data, ref.global_arena.data)
    eng.close()


def test_runs_that_roll_back_or_inject_faults_keep_the_plain_order():
    from nanofed_amd.engine import FaultPlan

    eng = FederatedEngine(EngineConfig(**_PREFETCH_CFG, checkpoint_rounds=True), comm=Communicator())
    assert not eng._may_prefetch()
    eng = FederatedEngine(EngineConfig(**_PREFETCH_CFG), comm=Communicator(),
                          faults=FaultPlan(drop={1: {0}}))
    assert not eng._may_prefetch()
    eng = FederatedEngine(EngineConfig(**_PREFETCH_CFG, sign_updates=True), comm=Communicator())
    assert not eng._may_prefetch()
    assert FederatedEngine(EngineConfig(**_PREFETCH_CFG), comm=Communicator())._may_prefetch()


def test_a_new_global_model_drops_the_round_that_was_queued_from_the_old_one():
    eng = FederatedEngine(EngineConfig(**_PREFETCH_CFG), comm=Communicator())
    eng.run_round(prefetch_next=True)
    assert eng._prelaunched is not None
    fresh = FederatedEngine(EngineConfig(**_PREFETCH_CFG, seed=5), comm=Communicator())
    eng.load_global_state_dict(fresh.global_state_dict())
    assert eng._prelaunched is None
    eng.run_round()
    # the same as a plain engine that is given that model after its first round
    ref = FederatedEngine(EngineConfig(**_PREFETCH_CFG, prefetch_rounds=False), comm=Communicator())
    ref.run_round()
    ref.load_global_state_dict(fresh.global_state_dict())
    ref.run_round()
    assert torch.equal(eng.global_arena.data, ref.global_arena.data)
