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


def test_pair_seeds_come_from_key_agreement_and_stay_with_the_clients():
    """Every client announces an X25519 public key; a pair's seed is what both ends derive
    from their own private key and the other's public key.  The engine object keeps the
    public keys and, per local client, that client's seeds - no table of all pairs."""
    from nanofed_amd.crypto.backend import X25519Key

    eng = FederatedEngine(cfg(num_clients=4, secure_agg="pairwise"), comm=Communicator())
    assert len(set(eng.secure_public_keys)) == 4 and all(len(k) == 32 for k in eng.secure_public_keys)
    assert not hasattr(eng, "_secure_seed_table") and not hasattr(eng, "_secure_seeds")
    rows = eng._own_seed_rows
    assert torch.equal(rows, rows.t()) and int(rows.diagonal().abs().sum()) == 0
    assert len({int(v) for v in rows[rows > 0]}) == 6  # one seed per pair, all different
    # fresh keys per engine: another engine of the same configuration shares no seed
    other = FederatedEngine(cfg(num_clients=4, secure_agg="pairwise"), comm=Communicator())
    assert set(other.secure_public_keys).isdisjoint(eng.secure_public_keys)
    # secure_key_seed: reproducible keys (tests, debugging)
    a = FederatedEngine(cfg(num_clients=4, secure_agg="pairwise", secure_key_seed=7), comm=Communicator())
    b = FederatedEngine(cfg(num_clients=4, secure_agg="pairwise", secure_key_seed=7), comm=Communicator())
    assert a.secure_public_keys == b.secure_public_keys and torch.equal(a._own_seed_rows, b._own_seed_rows)
    # the seed of pair (1, 2) is derivable from client 1's private key and 2's public key
    import hashlib

    sk1 = X25519Key(hashlib.sha256(b"nanofed-x25519|7|1").digest())
    digest = sk1.derive(a.secure_public_keys[2], b"nanofed-pair-mask|1|2")
    assert int(a._own_seed_rows[1, 2]) == int.from_bytes(digest[:8], "big") >> 1
    a.run_round()
    b.run_round()
    assert torch.equal(a.global_arena.data, b.global_arena.data)


def test_unknown_secure_mode_is_rejected():
    from nanofed_amd.core.exceptions import DeviceError

    with pytest.raises(DeviceError):
        FederatedEngine(cfg(secure_agg="xor"), comm=Communicator())
