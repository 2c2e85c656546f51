"""Security features on the engine's round (CPU): the z-score anomaly screen
turns a scaled update into a zero of the participation mask, and a round under
pairwise masking equals the plain round."""

import pytest
import torch

from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.parallel import Communicator


def cfg(**kw):
    base = dict(model="mnist_mlp", num_clients=8, samples_per_client=64, batch_size=32,
                local_epochs=1, eval_samples=100, dtype="fp32", backend="torch",
                model_kwargs={"hidden": 16}, learning_rate=0.05)
    base.update(kw)
    return EngineConfig(**base)


def test_a_fifty_fold_update_is_masked_out_by_the_z_score_screen():
    # reference capability: DefaultModelValidator.validate_statistics
    # (nanofed/server/validation.py:103-135), here wired into the round
    faults = FaultPlan(scale={0: {5: 50.0}})
    eng = FederatedEngine(cfg(zscore_threshold=2.0, min_clients_for_stats=5),
                          comm=Communicator(), faults=faults)
    rep = eng.run_round()
    valid = [c["valid"] for c in rep.client_metrics]
    assert valid == [True] * 5 + [False] + [True] * 2
    assert rep.client_metrics[5]["weight"] == 0.0
    assert rep.agg_metrics["anomalous_clients"] == 1.0
    norms = [c["update_norm"] for c in rep.client_metrics]
    assert norms[5] > 20 * max(norms[:5])
    # the aggregate is the weighted mean of the seven honest clients
    w = [c["weight"] for c in rep.client_metrics]
    want = sum(wk * row for wk, row in zip(w, eng.pool.rows()) if wk > 0)
    torch.testing.assert_close(eng.global_arena.data, want, rtol=1e-5, atol=1e-6)
    # without the screen the poisoned update moves the model far away
    loose = FederatedEngine(cfg(), comm=Communicator(), faults=FaultPlan(scale={0: {5: 50.0}}))
    loose.run_round()
    assert (loose.global_arena.data - want).norm() > 10 * (eng.global_arena.data - want).norm() + 1e-3


@pytest.mark.parametrize("dropped", [set(), {2, 6}])
def test_pairwise_masked_round_equals_the_plain_round(dropped):
    sizes = [64, 96, 64, 32, 64, 128, 64, 64]
    faults = lambda: FaultPlan(drop={0: set(dropped), 1: set(dropped)})  # noqa: E731
    plain = FederatedEngine(cfg(samples_per_client=sizes), comm=Communicator(), faults=faults())
    masked = FederatedEngine(cfg(samples_per_client=sizes, secure_agg="pairwise"),
                             comm=Communicator(), faults=faults())
    for _ in range(2):
        a, b = plain.run_round(), masked.run_round()
        assert a.status == b.status == "COMPLETED"
        assert a.num_clients == b.num_clients == 8 - len(dropped)
        torch.testing.assert_close(masked.global_arena.data, plain.global_arena.data,
                                   rtol=0, atol=1e-5)
    # what left the clients was blinded: no submitted row equals a trained model
    diffs = [(m - p).abs().mean().item() for m, p in zip(masked.pool.rows(), plain.pool.rows())]
    assert min(diffs) > 0.1


def test_unknown_secure_mode_is_rejected():
    from nanofed_amd.core.exceptions import DeviceError

    with pytest.raises(DeviceError):
        FederatedEngine(cfg(secure_agg="xor"), comm=Communicator())
