"""Native MNIST trainer at batch sizes other than 64: smaller batches are one partly
filled pass of the kernels, larger ones several passes whose gradients are added up
before one SGD update.  One local epoch (dropout off) is compared with the same SGD
steps of the fp32 PyTorch model on the same batches in the same order."""

import pytest
import torch
import torch.nn.functional as F

from nanofed_amd.engine import DPConfig, EngineConfig, FederatedEngine
from nanofed_amd.engine.mnist_native import micro_batches, supports
from nanofed_amd.models import MNISTModel
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu


def _engine(bs, n=300, clients=2, graph=True, lr=0.05, **kw):
    cfg = EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=n, batch_size=bs,
                       local_epochs=1, learning_rate=lr, eval_samples=128,
                       graph_capture=graph, seed=11, **kw)
    eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
    return eng


def test_which_configurations_take_the_native_trainer():
    assert [micro_batches(b) for b in (1, 32, 64, 65, 128, 256)] == [1, 1, 1, 2, 2, 4]
    for bs in (16, 32, 64, 100, 128, 256):
        assert supports(EngineConfig(model="mnist_cnn", batch_size=bs))
    assert not supports(EngineConfig(model="mnist_cnn", batch_size=2048))
    assert not supports(EngineConfig(model="mnist_cnn", batch_size=128, dp=DPConfig()))
    assert supports(EngineConfig(model="mnist_cnn", batch_size=64, dp=DPConfig()))
    assert _engine(128).backend == "native" and _engine(32).backend == "native"


@pytest.mark.parametrize("bs", [32, 100, 128, 256])
@pytest.mark.parametrize("graph", [True, False])
def test_one_epoch_follows_fp32_sgd_on_the_same_batches(bs, graph):
    n = 3 * bs - bs // 3  # three steps, the last batch partly filled
    eng = _engine(bs, n=n, graph=graph)
    trainer = eng.trainer
    for grp in [trainer.buf, *trainer._groups]:
        grp.ctx.p_drop1 = grp.ctx.p_drop2 = 0.0
    start = eng.global_arena.state_dict("cuda")
    report = eng.run_round()
    assert report.status == "COMPLETED"
    steps = 3
    for k in range(2):
        model = MNISTModel().cuda()
        model.load_state_dict(start)
        model.eval()  # dropout off
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        perm = eng.shards.epoch_permutation(k, 0, 0)
        losses, hits = [], []
        for s in range(steps):
            rows = perm[s * bs : (s + 1) * bs]
            x, y = eng.shards.batch(rows)
            opt.zero_grad()
            out = model(x)
            loss = F.nll_loss(out, y)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            hits.append(float((out.argmax(1) == y).float().mean()))
        got = eng.pool[k].state_dict("cuda")
        for name, p in model.named_parameters():
            want_delta = (p.detach() - start[name]).flatten()
            got_delta = (got[name] - start[name]).flatten()
            cos = F.cosine_similarity(got_delta, want_delta, dim=0)
            # several sequential steps: bf16 and fp32 trajectories drift apart slowly
            assert float(cos) > 0.95, (bs, name, float(cos))
            assert float(got_delta.norm()) == pytest.approx(float(want_delta.norm()), rel=0.08)
        m = report.client_metrics[k]["metrics"]
        assert m["loss"] == pytest.approx(sum(losses) / steps, rel=0.03)
        assert m["accuracy"] == pytest.approx(sum(hits) / steps, abs=0.03)
        assert m["samples_processed"] == n
    eng.close()


@pytest.mark.parametrize("bs", [32, 100, 128, 256])
def test_one_step_is_the_mean_gradient_of_the_whole_batch(bs):
    """One batch, one step: the update must be lr times the gradient of the mean loss
    over all `bs` samples taken at the SAME weights (accumulated passes, not sequential
    steps; a partly filled pass must not count its padding rows)."""
    eng = _engine(bs, n=bs, clients=1)
    for grp in [eng.trainer.buf, *eng.trainer._groups]:
        grp.ctx.p_drop1 = grp.ctx.p_drop2 = 0.0
    start = eng.global_arena.state_dict("cuda")
    eng.run_round()
    model = MNISTModel().cuda()
    model.load_state_dict(start)
    model.eval()
    perm = eng.shards.epoch_permutation(0, 0, 0)
    x, y = eng.shards.batch(perm)
    F.nll_loss(model(x), y).backward()
    got = eng.pool[0].state_dict("cuda")
    for name, p in model.named_parameters():
        step = ((start[name] - got[name]) / 0.05).flatten()
        cos = F.cosine_similarity(step, p.grad.flatten(), dim=0)
        assert float(cos) > 0.995, (name, float(cos))
        assert float(step.norm()) == pytest.approx(float(p.grad.norm()), rel=0.05)
    eng.close()


def test_rounds_are_reproducible_and_learn_at_batch_128():
    a, b = (_engine(128, n=2048, clients=4, lr=0.1) for _ in range(2))
    ra, rb = a.run(4), b.run(4)
    assert torch.equal(a.global_arena.data, b.global_arena.data)
    assert [r.agg_metrics for r in ra] == [r.agg_metrics for r in rb]
    assert ra[-1].agg_metrics["loss"] < 0.5 * ra[0].agg_metrics["loss"]
    assert a.evaluate()["accuracy"] > 0.8
    a.close()
    b.close()
