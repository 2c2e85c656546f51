"""MNIST federated experiment, three ways.

    python examples/mnist/run_experiment.py http      # the reference's shape: 1 server + 3 clients over localhost HTTP
    python examples/mnist/run_experiment.py innode    # same client code, in-node transport (no sockets / JSON)
    python examples/mnist/run_experiment.py device    # the device engine: clients are arena rows on the GPU(s)

All three run anywhere (CPU or GPU).  On a MI355X the clients' ``train_epoch``
calls of the first two modes and the rounds of the third run on the hand-written
HIP kernels (``NANOFED_NATIVE_TRAINER=0`` keeps ``train_epoch`` on PyTorch ops).  MNIST is read from ``./data/MNIST/raw`` if
the IDX files are there, otherwise a synthetic MNIST-shaped set is used.
"""

from __future__ import annotations

import asyncio
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))

from nanofed_amd import (  # noqa: E402
    Coordinator, CoordinatorConfig, FedAvgAggregator, HTTPClient, HTTPServer, ModelManager,
    TorchTrainer, TrainingConfig, coordinate,
)
from nanofed_amd.communication import InNodeClient, InNodeServer  # noqa: E402
from nanofed_amd.core.exceptions import NanoFedError  # noqa: E402
from nanofed_amd.data import load_mnist_data  # noqa: E402
from nanofed_amd.models import MNISTModel  # noqa: E402

CLIENTS = {"client_1": 12000, "client_2": 8000, "client_3": 4000}
ROUNDS, LOCAL_EPOCHS, LR = 2, 2, 0.1


async def run_client(make_client, client_id: str, num_samples: int, device: str) -> None:
    loader = load_mnist_data("./data", batch_size=64, train=True, download=False,
                             subset_fraction=num_samples / 60000, synthetic_fallback=True)
    trainer = TorchTrainer(TrainingConfig(epochs=LOCAL_EPOCHS, batch_size=64,
                                          learning_rate=LR, device=device))
    done: set[int] = set()
    async with make_client(client_id) as client:
        while not await client.check_server_status():
            try:
                state, rnd = await client.fetch_global_model()
            except NanoFedError:
                if await client.check_server_status():
                    break  # the last round closed between the two calls
                raise
            if rnd in done:
                await asyncio.sleep(0.05)
                continue
            model = MNISTModel().to(device)
            model.load_state_dict(state)
            optimizer = torch.optim.SGD(model.parameters(), lr=LR)
            for epoch in range(LOCAL_EPOCHS):
                metrics = await asyncio.to_thread(trainer.train_epoch, model, loader, optimizer, epoch)
            if await client.submit_update(model, metrics):
                done.add(rnd)


async def run_transport(kind: str) -> None:
    device = "cuda" if torch.cuda.is_available() else "cpu"
    server = HTTPServer("127.0.0.1", 8080) if kind == "http" else InNodeServer("example")
    await server.start()
    try:
        coordinator = Coordinator(
            ModelManager(MNISTModel().to(device)), FedAvgAggregator(), server,
            CoordinatorConfig(num_rounds=ROUNDS, min_clients=3, min_completion_rate=1.0,
                              round_timeout=600, base_dir=Path("runs/mnist")),
        )
        make = (lambda cid: HTTPClient(server.url, cid)) if kind == "http" else \
               (lambda cid: InNodeClient(server, cid))
        await asyncio.gather(
            coordinate(coordinator),
            *(run_client(make, cid, n, device) for cid, n in CLIENTS.items()),
        )
        print("global metrics:", coordinator.training_progress["global_metrics"])
    finally:
        await server.stop()


async def run_device() -> None:
    """The coordinator's calls, served by the device engine: same configs, same
    ``async for metrics in coordinator.start_training()``, same files under
    ``base_dir``; the clients are arena rows that train on this process's GPU."""
    from nanofed_amd import DeviceCoordinator
    from nanofed_amd.parallel import Communicator

    comm = Communicator.from_env()
    coordinator = DeviceCoordinator.create(
        CoordinatorConfig(num_rounds=ROUNDS, min_clients=len(CLIENTS), min_completion_rate=1.0,
                          round_timeout=600, base_dir=Path("runs/mnist_device")),
        TrainingConfig(epochs=LOCAL_EPOCHS, batch_size=64, learning_rate=LR),
        comm=comm, model="mnist_cnn", dataset="mnist", samples_per_client=list(CLIENTS.values()),
    )
    async for metrics in coordinator.start_training():
        print(f"round {metrics.round_id}: {metrics.status.name} {metrics.agg_metrics}")
    print("global metrics:", coordinator.training_progress["global_metrics"])
    print("held-out:", coordinator.engine.evaluate())
    coordinator.close()
    comm.close()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "device"
    if mode == "device":
        asyncio.run(run_device())
    else:
        asyncio.run(run_transport(mode))
