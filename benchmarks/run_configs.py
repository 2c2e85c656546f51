"""The five benchmark configurations, one JSON line each.

    python benchmarks/run_configs.py mlp_http          # 1: 2 CPU clients, 1 round over localhost HTTP
    python benchmarks/run_configs.py mnist_cnn         # 2: MNIST CNN, 8 clients, IID, bf16 (same as bench.py)
    python benchmarks/run_configs.py mnist_cnn_dp      # 3: Dirichlet(0.1) shards + DP-SGD clip/noise
    python benchmarks/run_configs.py resnet18          # 4: CIFAR-10-shaped ResNet-18, 8 clients, bf16
    python benchmarks/run_configs.py gpt2_125m         # 5: GPT-2 124M, 8 clients (large-arena reduce)

Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N``.
Data is synthetic (no network), weights are random-init.
"""

from __future__ import annotations

import argparse
import asyncio
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def engine_config(name: str, clients: int, samples: int = 0):
    from nanofed_amd.engine import DPConfig, EngineConfig

    if name == "mnist_cnn":
        return EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=7500,
                            batch_size=64, local_epochs=2, learning_rate=0.1)
    if name == "mnist_cnn_dp":
        return EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=7500,
                            batch_size=64, local_epochs=2, learning_rate=0.1,
                            partition="dirichlet", dirichlet_alpha=0.1,
                            dp=DPConfig(max_gradient_norm=1.0, noise_multiplier=1.1))
    if name == "resnet18":
        # four patterns per class + 8 % random labels: held-out accuracy stays below 1.0.
        # --samples-per-client 6250 is CIFAR-10's 50,000 images over 8 clients.
        return EngineConfig(model="resnet18", dataset="synthetic_cifar10_hard", num_clients=clients,
                            samples_per_client=samples or 1024, batch_size=64, local_epochs=1,
                            learning_rate=0.05, eval_samples=2000)
    if name == "gpt2_125m":
        # the token chain lives on the first 2048 ids (the model keeps all 50257
        # columns): 131k training tokens per round are enough to see it learned
        return EngineConfig(model="gpt2_125m", dataset="synthetic_tokens", num_clients=clients,
                            samples_per_client=16, batch_size=4, local_epochs=1, seq_len=1024,
                            learning_rate=0.2, eval_samples=8, token_active_vocab=2048,
                            gpt2_lanes=int(os.environ.get("NANOFED_GPT2_LANES", "1")))
    raise SystemExit(f"unknown config {name}")


def run_engine(name: str, args) -> dict:
    from nanofed_amd.engine import FederatedEngine
    from nanofed_amd.parallel import Communicator

    comm = Communicator.from_env()
    cfg = engine_config(name, args.clients, args.samples_per_client)
    engine = FederatedEngine(cfg, comm=comm)
    for _ in range(args.warmup):
        engine.run_round()
    comm.barrier()
    engine.synchronize()
    t0 = time.perf_counter()
    # as FederatedEngine.run does: round i + 1 is queued while the host finishes round i
    reports = [engine.run_round(prefetch_next=i + 1 < args.rounds) for i in range(args.rounds)]
    engine.synchronize()
    comm.barrier()
    dt = comm.max_over_ranks(time.perf_counter() - t0)
    ev = engine.evaluate()
    out = {
        "config": name, "n_gpus": comm.world_size, "clients": args.clients,
        "samples_per_client": cfg.samples_per_client, "dataset": cfg.dataset,
        "rounds": args.rounds, "rounds_per_s": args.rounds / dt,
        "ms_per_round": 1000 * dt / args.rounds,
        "arena_mb": engine.layout.numel * 4 / 1e6,
        "trainer": type(engine.trainer).__name__,
        "trainer_info": dict(getattr(engine.trainer, "stream_report", None) or {}),
        "train_loss": reports[-1].agg_metrics["loss"],
        "train_accuracy": reports[-1].agg_metrics["accuracy"],
        "phases_ms": {k: round(sum(r.phases_ms.get(k, 0.0) for r in reports) / len(reports), 3)
                      for k in sorted({k for r in reports for k in r.phases_ms})},
        "eval": ev, "data": "synthetic", "dtype": cfg.dtype,
        "status": [r.status for r in reports],
    }
    comm.close()
    return out if comm.is_root else {}


def run_mlp_http(args) -> dict:
    """Config 1: the reference's own transport, 2 CPU clients, one round."""
    import socket

    from nanofed_amd import (Coordinator, CoordinatorConfig, FedAvgAggregator, HTTPClient,
                             HTTPServer, ModelManager, TorchTrainer, TrainingConfig, coordinate)
    from nanofed_amd.data import synthetic_mnist
    from nanofed_amd.data.mnist import normalize_mnist
    from nanofed_amd.models import MNISTMLP

    def loader(n, seed):
        x, y = synthetic_mnist(n, seed=seed)
        ds = torch.utils.data.TensorDataset(normalize_mnist(x), y)
        return torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True)

    async def client(cid, url, n, seed):
        trainer = TorchTrainer(TrainingConfig(epochs=1, batch_size=64, learning_rate=0.1))
        data = loader(n, seed)
        done = set()
        async with HTTPClient(url, cid) as c:
            while not await c.check_server_status():
                state, rnd = await c.fetch_global_model()
                if rnd in done:
                    await asyncio.sleep(0.01)
                    continue
                model = MNISTMLP()
                model.load_state_dict(state)
                opt = torch.optim.SGD(model.parameters(), lr=0.1)
                m = trainer.train_epoch(model, data, opt, 0)
                if await c.submit_update(model, m):
                    done.add(rnd)

    async def main():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        server = HTTPServer("127.0.0.1", port)
        await server.start()
        with tempfile.TemporaryDirectory() as base:
            coord = Coordinator(ModelManager(MNISTMLP()), FedAvgAggregator(), server,
                                CoordinatorConfig(1, 2, 1.0, 120, base))
            t0 = time.perf_counter()
            await asyncio.gather(coordinate(coord), client("c1", server.url, 6000, 1),
                                 client("c2", server.url, 4000, 2))
            dt = time.perf_counter() - t0
        await server.stop()
        return dt, coord

    dt, coord = asyncio.run(main())
    return {"config": "mlp_http", "n_gpus": 0, "clients": 2, "rounds": 1,
            "rounds_per_s": 1.0 / dt, "ms_per_round": 1000 * dt,
            "global_metrics": coord.training_progress["global_metrics"], "data": "synthetic"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--clients", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples-per-client", type=int, default=0,
                    help="resnet18: samples per client (default 1024; 6250 = all of CIFAR-10 over 8)")
    ap.add_argument("--verbose", action="store_true", help="keep the INFO/DEBUG log lines")
    a = ap.parse_args()
    if not a.verbose:
        from nanofed_amd.utils import LogConfig, Logger, LogLevel

        Logger().configure(LogConfig(level=LogLevel.WARNING))
    res = run_mlp_http(a) if a.config == "mlp_http" else run_engine(a.config, a)
    if res:
        print(json.dumps(res), flush=True)
