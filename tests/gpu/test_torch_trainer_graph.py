"""Whole-step hipGraph capture of the PyTorch trainer: many replays, with the
round's eager work between them, must reproduce the eager run."""

import pytest
import torch

from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu


def _run(graph: bool, rounds: int = 3, **kw):
    torch.cuda.set_device(0)
    cfg = EngineConfig(backend="torch", graph_capture=graph, num_clients=2, local_epochs=1,
                       batch_size=32, seed=3, **kw)
    eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
    reps = eng.run(rounds)
    return eng.global_arena.data.float().cpu(), [r.agg_metrics["loss"] for r in reps], eng


def test_mlp_fp32_graph_equals_eager():
    kw = dict(model="mnist_mlp", samples_per_client=160, eval_samples=64, dtype="fp32",
              model_kwargs={"hidden": 32}, learning_rate=0.1)
    a, la, _ = _run(False, **kw)
    b, lb, eng = _run(True, **kw)
    assert len(eng.trainer._graphs) == 2
    torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-5)
    assert lb == pytest.approx(la, rel=1e-4)


def test_resnet_bf16_graph_tracks_eager_and_learns():
    # 150 samples: four full batches replay the graph, the ragged fifth runs eagerly
    kw = dict(model="resnet18", dataset="synthetic_cifar10", samples_per_client=150,
              eval_samples=64, learning_rate=0.05)
    # bf16 convolutions may pick different algorithms between runs and SGD amplifies
    # the difference round by round: compare one round (8 replays + 2 eager steps)
    # loosely, then check that the graph-captured run keeps learning
    a, la, _ = _run(False, rounds=1, **kw)
    b, lb, eng = _run(True, rounds=1, **kw)
    assert len(eng.trainer._graphs) == 2
    assert lb == pytest.approx(la, rel=0.1)
    rel = float((a - b).norm() / a.norm())
    assert rel < 0.05, rel
    _, losses, _ = _run(True, rounds=4, **kw)
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0]


def test_client_streams_train_the_same_models():
    """Four clients on concurrent streams (two of them share a stream when
    NANOFED_CLIENT_STREAMS is lowered) against the same clients trained one after
    another: fp32 runs of a model without dropout (whose masks depend on the launch
    order) must agree closely."""
    kw = dict(model="mnist_mlp", model_kwargs={"hidden": 64}, backend="torch", num_clients=4,
              local_epochs=2, batch_size=32, samples_per_client=160, eval_samples=64,
              dtype="fp32", learning_rate=0.05, seed=5)
    outs = []
    for streams in (False, True):
        torch.cuda.set_device(0)
        eng = FederatedEngine(EngineConfig(client_streams=streams, **kw),
                              comm=Communicator(device=torch.device("cuda", 0)))
        reps = eng.run(2)
        assert (eng.trainer._streams is not None) == streams
        outs.append((eng.global_arena.data.float().cpu(), [r.agg_metrics["loss"] for r in reps]))
    (a, la), (b, lb) = outs
    torch.testing.assert_close(b, a, rtol=1e-3, atol=1e-4)
    assert lb == pytest.approx(la, rel=1e-3)


def test_resnet_streams_count_batchnorm_steps():
    kw = dict(model="resnet18", dataset="synthetic_cifar10", backend="torch", num_clients=4,
              local_epochs=1, batch_size=32, samples_per_client=96, eval_samples=64,
              learning_rate=0.05, seed=2)
    torch.cuda.set_device(0)
    eng = FederatedEngine(EngineConfig(**kw), comm=Communicator(device=torch.device("cuda", 0)))
    reps = eng.run(2)
    assert all(r.status == "COMPLETED" for r in reps)
    tr = eng.trainer
    assert tr._streams is not None and tr._direct
    # 3 steps per epoch, 2 rounds: the counters live in the arena rows (global value + the
    # round's steps) and are averaged like every other entry
    sd = eng.global_state_dict()
    counts = {int(v) for k, v in sd.items() if k.endswith("num_batches_tracked")}
    assert counts == {6}
    # ... and every weight leaf is a bf16 view of the shadow arena without a gradient
    for p, _ in tr._leaves[0]:
        assert p.grad is None
        assert p.dtype == (torch.bfloat16 if p.dim() >= 2 else torch.float32)


def test_resnet_graphs_replayed_on_concurrent_streams_share_no_scratch():
    """Every client's step graph is captured on the main stream and replayed on the
    client's stream, four at a time.  Scratch memory that two graphs shared (the split-K
    buffer of the convolution GEMMs once did) would be overwritten by a neighbour: the
    run on streams must give the model of the run that trains client after client."""
    kw = dict(model="resnet18", dataset="synthetic_cifar10", backend="torch", num_clients=4,
              local_epochs=1, batch_size=32, samples_per_client=128, eval_samples=64,
              learning_rate=0.05, seed=4)
    outs = []
    for streams in (False, True):
        torch.cuda.set_device(0)
        eng = FederatedEngine(EngineConfig(client_streams=streams, **kw),
                              comm=Communicator(device=torch.device("cuda", 0)))
        reps = eng.run(2)
        assert (eng.trainer._streams is not None) == streams
        assert len(eng.trainer._graphs) == 4
        outs.append((eng.global_arena.data.float().cpu(), reps[-1].agg_metrics["loss"]))
    (a, la), (b, lb) = outs
    rel = float((a - b).norm() / a.norm())
    assert rel < 1e-3, rel
    assert lb == pytest.approx(la, rel=1e-2)
