"""The reference-style calls on the hand-written kernels: ``TorchTrainer.train_epoch``
routed to the native MNIST step, and ``DeviceCoordinator`` rounds."""

import asyncio
import json

import numpy as np
import pytest
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils.data import DataLoader

from nanofed_amd import CoordinatorConfig, DeviceCoordinator, TrainingConfig
from nanofed_amd.data.mnist import MNISTTensorDataset
from nanofed_amd.data.synthetic import synthetic_mnist
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.engine.data import FederatedData
from nanofed_amd.engine.mnist_native import NativeBuffers
from nanofed_amd.models import MNISTModel
from nanofed_amd.parallel import Communicator
from nanofed_amd.trainer import TorchTrainer, TrainingConfig as TC

pytestmark = pytest.mark.gpu


class _OpLog(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.ops.append(str(func))
        return func(*args, **(kwargs or {}))


def _loader(n, seed=5):
    images, labels = synthetic_mnist(n, seed=seed)
    torch.manual_seed(0)
    return DataLoader(MNISTTensorDataset(images, labels), batch_size=64, shuffle=True), images, labels


def test_train_epoch_runs_the_hand_written_kernels(monkeypatch):
    loader, _, _ = _loader(1000)
    trainer = TorchTrainer(TC(epochs=3, batch_size=64, learning_rate=0.05, device="cuda"))
    torch.manual_seed(1)
    model = MNISTModel().to("cuda")
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    calls = []
    real = NativeBuffers.train_steps
    monkeypatch.setattr(NativeBuffers, "train_steps",
                        lambda self, first, count: (calls.append(count), real(self, first, count))[1])
    first = trainer.train_epoch(model, loader, opt, 0)   # builds the session, captures the graph
    assert sum(calls) >= 16  # warm-up + capture of the 16 steps of an epoch
    with _OpLog() as log:
        second = trainer.train_epoch(model, loader, opt, 1)
        third = trainer.train_epoch(model, loader, opt, 2)
    heavy = [op for op in log.ops if any(s in op for s in
             ("convolution", "aten.mm", "aten.addmm", "max_pool", "log_softmax", "nll_loss"))]
    assert heavy == [], f"PyTorch compute ops ran inside the native route: {heavy}"
    assert len(trainer._native_sessions) == 1
    assert first.samples_processed == 1000 and first.batch == 15 and third.epoch == 2
    assert np.isfinite([first.loss, second.loss, third.loss]).all()
    assert third.loss < first.loss and third.accuracy > first.accuracy
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_native_and_pytorch_paths_learn_the_same_task(monkeypatch):
    def run(native_on):
        monkeypatch.setenv("NANOFED_NATIVE_TRAINER", "1" if native_on else "0")
        loader, _, _ = _loader(2048)
        trainer = TorchTrainer(TC(epochs=6, batch_size=64, learning_rate=0.05, device="cuda"))
        torch.manual_seed(1)
        model = MNISTModel().to("cuda")
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        out = [trainer.train_epoch(model, loader, opt, e) for e in range(6)]
        assert ("_native_sessions" in trainer.__dict__) == native_on
        return out

    a, b = run(True), run(False)
    # Different shuffles and dropout streams, bf16 against fp32: the curves agree in the
    # large, not in the digits.  Compared where both have flattened (the loss falls from
    # 1.5 to 0.3 within two epochs, and which epoch that is depends on the shuffle:
    # tools/probes/route_curve_probe.py prints the curves of three seeds).
    assert a[0].loss > 1.5 and b[0].loss > 1.5
    assert a[-1].loss < 0.25 and b[-1].loss < 0.25
    assert a[-1].loss == pytest.approx(b[-1].loss, abs=0.08)
    assert a[-1].accuracy == pytest.approx(b[-1].accuracy, abs=0.04)


def test_one_epoch_equals_one_engine_round_of_one_client():
    n = 640
    loader, images, labels = _loader(n)
    ds = loader.dataset
    data = FederatedData("image", images, labels, [np.arange(n)], None, None,
                         (float(ds.mean),), (float(ds.std),))
    engine = FederatedEngine(
        EngineConfig(model="mnist_cnn", num_clients=1, samples_per_client=n, local_epochs=1,
                     batch_size=64, learning_rate=0.05, eval_samples=0, backend="native"),
        comm=Communicator(device=torch.device("cuda", 0)), data=data,
    )
    model = MNISTModel().to("cuda")
    model.load_state_dict(engine.global_arena.state_dict("cuda"))
    trainer = TorchTrainer(TC(epochs=1, batch_size=64, learning_rate=0.05, device="cuda"))
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    m = trainer.train_epoch(model, loader, opt, 0)
    report = engine.run_round()
    want = engine.global_arena.state_dict("cuda")
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v, want[k], rtol=0, atol=0)
    assert m.loss == pytest.approx(report.agg_metrics["loss"], rel=1e-6)
    engine.close()


def test_device_coordinator_rounds_on_the_native_trainer(tmp_path):
    config = CoordinatorConfig(num_rounds=2, min_clients=8, min_completion_rate=1.0,
                               round_timeout=300, base_dir=tmp_path / "run")
    kw = dict(model="mnist_cnn", samples_per_client=512, eval_samples=256)
    coordinator = DeviceCoordinator.create(
        config, TrainingConfig(epochs=2, batch_size=64, learning_rate=0.1),
        comm=Communicator(device=torch.device("cuda", 0)), **kw,
    )
    assert coordinator.engine.backend == "native"

    async def main():
        return [m async for m in coordinator.start_training()]

    rounds = asyncio.run(main())
    assert [m.round_id for m in rounds] == [0, 1] and rounds[1].num_clients == 8
    assert rounds[1].agg_metrics["loss"] < rounds[0].agg_metrics["loss"]
    record = json.loads((config.base_dir / "metrics" / "metrics_round_1.json").read_text())
    assert record["status"] == "COMPLETED" and len(record["client_metrics"]) == 8
    versions = coordinator.model_manager.list_versions()
    assert len(versions) == 3
    saved = torch.load(versions[-1].path, weights_only=True)
    live = coordinator.engine.global_state_dict()
    for k in saved:
        torch.testing.assert_close(saved[k], live[k], rtol=0, atol=0)
    # the same engine without the coordinator: identical model
    plain = FederatedEngine(coordinator.engine.cfg, comm=Communicator(device=torch.device("cuda", 0)))
    plain.run(2)
    torch.testing.assert_close(plain.global_arena.data, coordinator.engine.global_arena.data,
                               rtol=0, atol=0)
    coordinator.close()
    plain.close()


def test_mixed_round_on_the_native_trainer_with_rounds_on_the_worker_thread(tmp_path):
    """Six resident clients on the hand-written kernels + two clients that submit through
    the coordinator's in-node server, the round itself on the coordinator's worker thread
    (a server is attached): bit-identical to the eight-client engine, and the event loop
    gets its turns while the device trains."""
    from dataclasses import replace

    from nanofed_amd.communication.innode import InNodeClient, InNodeServer
    from nanofed_amd.engine import EngineConfig
    from nanofed_amd.engine.data import build_federated_data

    torch.cuda.set_device(0)
    base = dict(model="mnist_cnn", samples_per_client=512, batch_size=64, local_epochs=1,
                eval_samples=256, learning_rate=0.1, placement_invariant=True, seed=3)
    full_cfg = EngineConfig(num_clients=8, **base)
    data = build_federated_data(full_cfg)
    ref = FederatedEngine(full_cfg, comm=Communicator(device=torch.device("cuda", 0)), data=data)
    ref.run_round()
    torch.cuda.synchronize()

    async def main():
        server = InNodeServer("mixed-gpu")
        await server.start()
        coord = DeviceCoordinator.create(
            CoordinatorConfig(num_rounds=1, min_clients=6, min_completion_rate=1.0,
                              round_timeout=60, base_dir=tmp_path / "run"),
            server=server, comm=Communicator(device=torch.device("cuda", 0)),
            data=replace(data, shards=data.shards[:6]), external_updates=2, external_slots=2,
            **base)
        assert coord.engine.backend == "native" and coord._background
        ticks = []
        done = asyncio.Event()

        async def outsider(k):
            async with InNodeClient(server, f"outside_{k}") as client:
                _, rnd = await client.fetch_global_model()
                assert rnd == 0
                assert await client.submit_update(ref.pool[k].state_dict("cpu"),
                                                  {"loss": 1.0, "accuracy": 0.5, "num_samples": 512})

        async def poller():
            while not done.is_set():
                ticks.append(server.status()["num_updates"])
                await asyncio.sleep(0.001)

        async def rounds():
            try:
                return [m async for m in coord.start_training()]
            finally:
                done.set()

        _, _, _, metrics = await asyncio.gather(outsider(6), outsider(7), poller(), rounds())
        await server.stop()
        return coord, metrics, ticks

    coord, metrics, ticks = asyncio.run(main())
    assert metrics[0].num_clients == 8 and metrics[0].status.name == "COMPLETED"
    assert torch.equal(coord.engine.global_arena.data, ref.global_arena.data)
    assert len(ticks) >= 2
    coord.close()
    ref.close()
