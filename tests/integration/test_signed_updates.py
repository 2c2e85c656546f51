"""Signed updates on the device plane: every simulated client signs the tree digest of its
arena row; an update altered after signing is left out of the round."""

import hashlib

import pytest
import torch

from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.ops import digest_ops
from nanofed_amd.parallel import Communicator


def _engine(**kw):
    cfg = dict(model="mnist_mlp", model_kwargs={"hidden": 32}, num_clients=3, samples_per_client=64,
               batch_size=32, local_epochs=1, eval_samples=64, learning_rate=0.05, seed=3,
               backend="torch", dtype="fp32", sign_updates=True)
    cfg.update(kw)
    faults = cfg.pop("faults", None)
    return FederatedEngine(EngineConfig(**cfg), comm=Communicator(device=torch.device("cpu")),
                           faults=faults)


def test_row_digest_is_a_sha256_tree_over_the_row_bytes():
    torch.manual_seed(0)
    rows = torch.randn(3, 1024 + 256)
    got = digest_ops.row_digests(rows, chunk_bytes=1024)
    raw = rows[1].numpy().tobytes()
    leaves = b"".join(hashlib.sha256(raw[i : i + 1024]).digest() for i in range(0, len(raw), 1024))
    head = b"nanofed-amd/row-digest/v1" + len(raw).to_bytes(8, "big") + (1024).to_bytes(8, "big")
    assert got[1] == hashlib.sha256(head + leaves).digest()
    assert len(set(got)) == 3
    # one changed value changes the digest; strided rows hash like contiguous ones
    other = rows.clone()
    other[1, 700] += 1e-3
    assert digest_ops.row_digests(other, 1024)[1] != got[1]
    wide = torch.zeros(3, 2048)
    wide[:, :1280] = rows
    assert digest_ops.row_digests(wide[:, :1280], 1024) == got


def test_honest_round_verifies_every_client():
    eng = _engine()
    assert len(eng.client_public_keys) == 3
    reps = eng.run(2)
    assert [r.status for r in reps] == ["COMPLETED", "COMPLETED"]
    assert all(r.num_clients == 3 for r in reps) and eng.unverified_updates == 0
    plain = _engine(sign_updates=False)
    plain.run(2)
    assert torch.equal(eng.global_arena.data, plain.global_arena.data)  # signing changes nothing


def test_update_altered_after_signing_is_left_out():
    eng = _engine(faults=FaultPlan(tamper={1: {2}}))
    reps = eng.run(2)
    assert reps[0].num_clients == 3
    assert reps[1].status == "COMPLETED" and reps[1].num_clients == 2
    assert not reps[1].client_metrics[2]["valid"] and reps[1].client_metrics[2]["weight"] == 0.0
    assert eng.unverified_updates == 1
    # without signatures the altered update is averaged in
    blind = _engine(sign_updates=False, faults=FaultPlan(tamper={1: {2}}))
    blind_reps = blind.run(2)
    assert blind_reps[1].num_clients == 3
    assert not torch.equal(eng.global_arena.data, blind.global_arena.data)


def test_signature_binds_round_and_client():
    eng = _engine()
    eng.pool.broadcast_from(eng.global_arena.data)
    sigs = eng._sign_rows(0)
    local = torch.ones(3, 6)
    eng._verify_rows(0, sigs, local)
    assert local[:, 3].tolist() == [1.0, 1.0, 1.0]
    replayed = torch.ones(3, 6)
    eng._verify_rows(1, sigs, replayed)            # signatures of round 0 presented in round 1
    assert replayed[:, 3].tolist() == [0.0, 0.0, 0.0]
    swapped = torch.ones(3, 6)
    eng._verify_rows(0, [sigs[1], sigs[0], sigs[2]], swapped)   # another client's signature
    assert swapped[:, 3].tolist() == [0.0, 0.0, 1.0]


@pytest.mark.gpu
def test_device_digests_equal_the_host_tree():
    torch.manual_seed(1)
    rows = torch.randn(4, 1408, device="cuda")   # 5632 bytes: five and a half 1 KiB leaves
    pool = torch.zeros(4, 2048, device="cuda")
    pool[:, : rows.shape[1]] = rows
    for chunk in (1024, 4096, 64):
        assert digest_ops.row_digests(rows, chunk) == digest_ops.row_digests(rows.cpu(), chunk)
        # rows of a wider pool (strided)
        assert digest_ops.row_digests(pool[:, : rows.shape[1]], chunk) == digest_ops.row_digests(rows.cpu(), chunk)
    leaves = digest_ops.chunk_digests(rows, 1024)
    assert leaves.shape == (4, 6, 32) and leaves.dtype == torch.uint8
    raw = rows[2].cpu().numpy().tobytes()
    assert bytes(leaves[2, 5].cpu().tolist()) == hashlib.sha256(raw[5 * 1024 :]).digest()


@pytest.mark.gpu
def test_signed_native_mnist_round_on_the_gpu():
    torch.cuda.set_device(0)
    cfg = EngineConfig(model="mnist_cnn", num_clients=4, samples_per_client=256, batch_size=64,
                       local_epochs=1, eval_samples=256, sign_updates=True, seed=2)
    eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)),
                          faults=FaultPlan(tamper={1: {0}}))
    reps = eng.run(3)
    assert [r.num_clients for r in reps] == [4, 3, 4]
    assert eng.unverified_updates == 1 and "verify_signatures" in reps[0].phases_ms
