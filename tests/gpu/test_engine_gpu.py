"""End-to-end federated rounds on a real GPU."""

import pytest
import torch

from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu


def _engine(**kw):
    torch.cuda.set_device(0)
    comm = Communicator(device=torch.device("cuda", 0))
    base = dict(model="mnist_cnn", num_clients=4, samples_per_client=256, batch_size=64,
                local_epochs=1, eval_samples=512, backend="torch")
    base.update(kw)
    return FederatedEngine(EngineConfig(**base), comm=comm)


def test_torch_backend_learns_on_gpu():
    eng = _engine(local_epochs=2, samples_per_client=512)
    reports = eng.run(3)
    assert all(r.status == "COMPLETED" for r in reports)
    assert reports[-1].agg_metrics["loss"] < reports[0].agg_metrics["loss"]
    ev = eng.evaluate()
    assert ev["accuracy"] > 0.5


def test_poisoned_client_is_masked_out():
    eng = _engine()
    eng.faults = FaultPlan(poison={0: {1}})
    rep = eng.run_round()
    assert rep.status == "COMPLETED" and rep.num_clients == 3
    assert not rep.client_metrics[1]["valid"]
    assert torch.isfinite(eng.global_arena.data).all()


def test_resnet18_round_with_batchnorm_buffers():
    eng = _engine(model="resnet18", dataset="synthetic_cifar10", num_clients=2,
                  samples_per_client=128, batch_size=32, eval_samples=64,
                  model_kwargs={"width": 16}, learning_rate=0.05)
    rep = eng.run_round()
    assert rep.status == "COMPLETED"
    sd = eng.global_state_dict()
    assert sd["bn1.num_batches_tracked"].dtype == torch.int64
    assert int(sd["bn1.num_batches_tracked"]) == 4
    assert torch.isfinite(sd["bn1.running_var"]).all()
    assert "accuracy" in eng.evaluate()


def test_small_gpt2_round_uses_one_flat_arena():
    eng = _engine(model="gpt2_125m", dataset="synthetic_tokens", num_clients=2,
                  samples_per_client=8, batch_size=4, seq_len=64, eval_samples=4,
                  model_kwargs=dict(vocab_size=512, n_positions=64, n_embd=64, n_layer=2, n_head=4),
                  learning_rate=0.05)
    r0 = eng.run_round()
    r1 = eng.run_round()
    assert r0.status == r1.status == "COMPLETED"
    assert r1.agg_metrics["loss"] < r0.agg_metrics["loss"] + 0.5
    assert eng.evaluate()["loss"] > 0


@pytest.mark.parametrize("backend,clients", [("auto", 8), ("auto", 4), ("auto", 1)])
def test_rounds_queued_early_give_the_same_model_files_and_phases(tmp_path, backend, clients):
    """run() queues round r + 1 on the device before the host has read the outcome of round
    r: the models, the metrics and the files on disk must not depend on it.  (Checked on the
    native trainer, which is bit-reproducible; the library kernels of the PyTorch-op trainer
    are not, run to run, whatever the order.)"""
    import json

    def go(prefetch, base_dir):
        torch.cuda.set_device(0)
        eng = FederatedEngine(
            EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=640,
                         batch_size=64, local_epochs=2, eval_samples=256, backend=backend,
                         prefetch_rounds=prefetch),
            comm=Communicator(device=torch.device("cuda", 0)), base_dir=base_dir)
        reports = eng.run(4)
        eng.close()
        return eng, reports

    early, ra = go(True, tmp_path / "early")
    plain, rb = go(False, tmp_path / "plain")
    assert torch.equal(early.global_arena.data, plain.global_arena.data)
    assert [r.agg_metrics for r in ra] == [r.agg_metrics for r in rb]
    assert [r.status for r in ra] == ["COMPLETED"] * 4
    for rep in ra:
        assert {"broadcast", "local_train"} <= set(rep.phases_ms)
    # the version written for round r holds the model after round r, not a later one
    for r in range(4):
        a = json.loads((tmp_path / "early" / "metrics" / f"metrics_round_{r}.json").read_text())
        b = json.loads((tmp_path / "plain" / "metrics" / f"metrics_round_{r}.json").read_text())
        assert a["agg_metrics"] == b["agg_metrics"] and a["round_id"] == r
    va = sorted((tmp_path / "early" / "models").rglob("*.pt"))
    vb = sorted((tmp_path / "plain" / "models").rglob("*.pt"))
    assert len(va) == len(vb) >= 4  # the four rounds (+ the initial model)
    va, vb = va[-4:], vb[-4:]
    for fa, fb in zip(va, vb):
        sa = torch.load(fa, weights_only=True)
        sb = torch.load(fb, weights_only=True)
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
    # consecutive versions differ (a snapshot taken too late would repeat the next one)
    s0, s1 = torch.load(va[0], weights_only=True), torch.load(va[1], weights_only=True)
    assert any(not torch.equal(s0[k], s1[k]) for k in s0)


def test_pytorch_op_trainer_with_rounds_queued_early_learns_and_writes_every_round(tmp_path):
    torch.cuda.set_device(0)
    eng = FederatedEngine(
        EngineConfig(model="mnist_cnn", num_clients=4, samples_per_client=640, batch_size=64,
                     local_epochs=2, eval_samples=256, backend="torch"),
        comm=Communicator(device=torch.device("cuda", 0)), base_dir=tmp_path)
    assert eng._may_prefetch()
    reports = eng.run(4)
    eng.close()
    assert [r.round_id for r in reports] == [0, 1, 2, 3]
    assert reports[-1].agg_metrics["loss"] < 0.5 * reports[0].agg_metrics["loss"]
    assert len(list((tmp_path / "metrics").glob("metrics_round_*.json"))) == 4
    versions = sorted((tmp_path / "models").rglob("*.pt"))[-4:]
    states = [torch.load(v, weights_only=True) for v in versions]
    for a, b in zip(states, states[1:]):
        assert any(not torch.equal(a[k], b[k]) for k in a)
    last = eng.global_state_dict()
    assert all(torch.equal(states[-1][k], last[k]) for k in last)
