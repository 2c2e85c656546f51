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
