"""Failure handling of the device-plane engine (CPU): injected mid-round
errors roll back to the last completed checkpoint, stragglers beyond the round
time-out are masked out, a finished run can be resumed, and every round
reports per-phase timings."""

import json

import pytest
import torch

from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.parallel import Communicator
from nanofed_amd.utils.trace import PhaseTimer, roctx_available, trace_mark, trace_range


def cfg(**kw):
    base = dict(model="mnist_mlp", num_clients=4, samples_per_client=96, batch_size=32,
                local_epochs=1, eval_samples=100, dtype="fp32", backend="torch",
                model_kwargs={"hidden": 16}, learning_rate=0.1)
    base.update(kw)
    return EngineConfig(**base)


def test_injected_runtime_error_rolls_back_to_last_completed_round(tmp_path):
    clean = FederatedEngine(cfg(), comm=Communicator())
    clean.run(3)

    faults = FaultPlan(fail={1: "HIP error: injected"})
    eng = FederatedEngine(cfg(checkpoint_rounds=True), comm=Communicator(),
                          base_dir=tmp_path, faults=faults)
    reports = eng.run(3)
    assert eng.recoveries == 1
    assert [r.round_id for r in reports] == [0, 1, 2]
    assert all(r.status == "COMPLETED" for r in reports)
    # round 1 restarted from round 0's checkpoint, so the run ends where a clean one does
    a, b = clean.global_state_dict(), eng.global_state_dict()
    assert all(torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-6) for k in a)
    # checkpoint layout of the reference's FileStateStore
    ck = tmp_path / "checkpoints" / "round_2"
    meta = json.loads((ck / "metadata.json").read_text())
    assert meta["state"] == "COMPLETED" and meta["round_id"] == 2
    state = torch.load(ck / "state.pt", weights_only=True)
    assert state["next_round"] == 3 and set(state["model_state"]) == set(a)


def test_failure_before_any_checkpoint_restarts_from_the_initial_model(tmp_path):
    faults = FaultPlan(fail={0: "RCCL error: injected"})
    eng = FederatedEngine(cfg(checkpoint_rounds=True), comm=Communicator(),
                          base_dir=tmp_path, faults=faults)
    reports = eng.run(1)
    assert eng.recoveries == 1 and reports[0].round_id == 0


def test_unrecoverable_or_unconfigured_failures_propagate(tmp_path):
    eng = FederatedEngine(cfg(), comm=Communicator(), faults=FaultPlan(fail={0: "boom"}))
    with pytest.raises(RuntimeError, match="boom"):
        eng.run(1)

    class Odd(FederatedEngine):
        def run_round(self):
            raise ValueError("not recoverable")

    odd = Odd(cfg(checkpoint_rounds=True), comm=Communicator(), base_dir=tmp_path)
    with pytest.raises(ValueError):
        odd.run(1)


def test_recovery_budget_is_bounded(tmp_path):
    class Flaky(FederatedEngine):
        def run_round(self):
            raise RuntimeError("always")

    eng = Flaky(cfg(checkpoint_rounds=True, max_recoveries=2), comm=Communicator(),
                base_dir=tmp_path)
    with pytest.raises(RuntimeError, match="always"):
        eng.run(1)
    assert eng.recoveries == 2


def test_stragglers_beyond_the_round_timeout_are_left_out():
    faults = FaultPlan(delay={0: {2: 5.0, 3: 0.1}})
    eng = FederatedEngine(cfg(round_timeout_s=1.0), comm=Communicator(), faults=faults)
    rep = eng.run_round()
    valid = [c["valid"] for c in rep.client_metrics]
    assert valid == [True, True, False, True]
    assert rep.client_metrics[2]["weight"] == 0.0


def test_resume_continues_the_round_counter(tmp_path):
    first = FederatedEngine(cfg(), comm=Communicator(), base_dir=tmp_path)
    first.run(2)
    want = first.global_state_dict()

    second = FederatedEngine(cfg(seed=7), comm=Communicator(), base_dir=tmp_path)
    assert second.resume() == 2
    got = second.global_state_dict()
    assert all(torch.equal(want[k], got[k]) for k in want)
    rep = second.run_round()
    assert rep.round_id == 2
    assert (tmp_path / "metrics" / "metrics_round_2.json").exists()


def test_round_report_carries_phase_timings(tmp_path):
    eng = FederatedEngine(cfg(), comm=Communicator(), base_dir=tmp_path)
    rep = eng.run_round()
    assert {"broadcast", "local_train", "screen_gather", "reduce", "checkpoint"} <= set(rep.phases_ms)
    assert all(v >= 0.0 for v in rep.phases_ms.values())
    saved = json.loads((tmp_path / "metrics" / "metrics_round_0.json").read_text())
    # the file is written inside the checkpoint phase, so it holds the device phases only
    assert set(saved["phases_ms"]) | {"checkpoint"} == set(rep.phases_ms)
    quiet = FederatedEngine(cfg(phase_timing=False), comm=Communicator())
    assert quiet.run_round().phases_ms == {}


def test_phase_timer_and_trace_ranges_on_cpu():
    timer = PhaseTimer("cpu")
    with timer.phase("a"):
        sum(range(1000))
    with timer.phase("a"):
        pass
    with timer.phase("b"), trace_range("inner"):
        trace_mark("mark")
    out = timer.collect()
    assert set(out) == {"a", "b"} and out["a"] >= 0.0
    assert timer.collect() == {}
    assert isinstance(roctx_available(), bool)


def test_a_failed_collective_at_the_deadline_is_the_time_out_not_a_traceback(monkeypatch):
    """When a peer's watchdog fires a moment before ours, its exit reaches this rank as a
    failed collective; the round must end as the time-out it is (FAILED record, supervisor
    exit code), not race the watchdog thread with an exception."""
    import time as _time

    class TimedOut(Exception):
        pass

    eng = FederatedEngine(cfg(round_timeout_s=0.6, enforce_round_timeout=True,
                              first_round_grace_s=0.0), comm=Communicator())
    seen = []

    def on_timeout(round_id, limit_s):
        seen.append((round_id, limit_s))
        raise TimedOut()

    monkeypatch.setattr(eng, "_on_round_timeout", on_timeout)
    eng._watchdog.close()  # the thread would end the process; the main-thread path is under test

    def peer_left(r, t0, start, prefetch_next=False):
        _time.sleep(0.5)  # 0.1 s before the deadline
        raise RuntimeError("Connection closed by peer")

    monkeypatch.setattr(eng, "_round", peer_left)
    with pytest.raises(TimedOut):
        eng.run_round()
    assert seen == [(0, 0.6)]

    # an error early in the window is an error
    def early(r, t0, start, prefetch_next=False):
        raise RuntimeError("boom")

    monkeypatch.setattr(eng, "_round", early)
    with pytest.raises(RuntimeError, match="boom"):
        eng.run_round()
    assert len(seen) == 1
    eng.close()
