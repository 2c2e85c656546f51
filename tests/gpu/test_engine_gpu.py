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
