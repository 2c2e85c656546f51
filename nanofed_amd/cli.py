"""``nanofed`` console entry point.

The reference declares ``nanofed = "nanofed.cli:main"`` (pyproject.toml:22-23)
but ships no ``cli`` module.  This one covers the common operations:

    nanofed build                       compile the native libraries in-tree
    nanofed info                        versions, devices, native-library status
    nanofed run [--model ...]           run federated rounds with the device engine
    nanofed bench [bench.py flags]      the headline benchmark
    nanofed serve --port P              start an HTTP server + coordinator
"""

from __future__ import annotations

import asyncio
import json
from pathlib import Path

import click


@click.group()
def main() -> None:
    """MI355X-native federated learning."""


@main.command()
@click.option("--force", is_flag=True, help="recompile everything")
def build(force: bool) -> None:
    from nanofed_amd import native_build

    for name, path in native_build.build_all(force=force).items():
        click.echo(f"{name}: {path}")


@main.command()
def info() -> None:
    import torch

    import nanofed_amd
    from nanofed_amd import native_build

    click.echo(f"nanofed_amd {nanofed_amd.__version__}, torch {torch.__version__}")
    click.echo(f"HIP devices: {torch.cuda.device_count()}")
    for lib in (native_build.HOST_LIB, native_build.HIP_LIB):
        click.echo(f"{lib.name}: {'present' if lib.exists() else 'missing (run `nanofed build`)'}")


@main.command()
@click.option("--model", default="mnist_cnn")
@click.option("--dataset", default=None)
@click.option("--clients", default=8)
@click.option("--rounds", default=2)
@click.option("--local-epochs", default=2)
@click.option("--batch-size", default=64)
@click.option("--samples-per-client", default=7500)
@click.option("--lr", default=0.1)
@click.option("--partition", type=click.Choice(["iid", "dirichlet"]), default="iid")
@click.option("--alpha", default=0.1, help="Dirichlet concentration")
@click.option("--dp-noise", default=0.0, help="DP-SGD noise multiplier (0 = off)")
@click.option("--dp-clip", default=1.0)
@click.option("--backend", type=click.Choice(["auto", "native", "torch"]), default="auto")
@click.option("--base-dir", default=None, type=click.Path(), help="write models/ and metrics/ here")
@click.option("--data-dir", default="./data", type=click.Path(),
              help="where --dataset mnist / cifar10 look for their files (IDX files; CIFAR-10 binary batches)")
@click.option("--seq-len", default=1024, help="sequence length of token datasets")
@click.option("--min-clients", default=1, help="quorum: valid updates needed to close a round")
@click.option("--max-update-norm", default=0.0, help="reject updates farther than this from the global model")
@click.option("--checkpoint-rounds", is_flag=True,
              help="checkpoint every completed round and roll back on recoverable failures")
@click.option("--resume", is_flag=True, help="continue from the newest model version in --base-dir")
@click.option("--phases", is_flag=True, help="print the HIP-event timing of the round phases")
@click.option("--zscore", default=0.0, help="leave out updates whose norm is this many sigmas off (0 = off)")
@click.option("--secure-agg", type=click.Choice(["none", "pairwise"]), default="none",
              help="blind client rows with pairwise masks before the reduce")
@click.option("--sign-updates", is_flag=True, help="clients sign their updates (RSA-PSS), verified before aggregation")
def run(model, dataset, clients, rounds, local_epochs, batch_size, samples_per_client, lr,
        partition, alpha, dp_noise, dp_clip, backend, base_dir, data_dir, seq_len, min_clients,
        max_update_norm, checkpoint_rounds, resume, phases, zscore, secure_agg,
        sign_updates) -> None:
    """Federated rounds on the device plane (launch under torch.distributed.run for multi-GPU)."""
    from nanofed_amd.engine import DPConfig, EngineConfig, FederatedEngine
    from nanofed_amd.parallel import Communicator

    default_ds = {"mnist_cnn": "synthetic_mnist", "mnist_mlp": "synthetic_mnist",
                  "resnet18": "synthetic_cifar10_hard", "gpt2_125m": "synthetic_tokens"}
    cfg = EngineConfig(
        model=model, dataset=dataset or default_ds.get(model, "synthetic_mnist"),
        num_clients=clients, num_rounds=rounds, local_epochs=local_epochs,
        batch_size=batch_size, samples_per_client=samples_per_client, learning_rate=lr,
        partition=partition, dirichlet_alpha=alpha, backend=backend, seq_len=seq_len,
        min_clients=min_clients, max_update_norm=max_update_norm,
        checkpoint_rounds=checkpoint_rounds, zscore_threshold=zscore, secure_agg=secure_agg,
        sign_updates=sign_updates, data_dir=str(data_dir),
        dp=DPConfig(max_gradient_norm=dp_clip, noise_multiplier=dp_noise) if dp_noise > 0 else None,
    )
    comm = Communicator.from_env()
    if (checkpoint_rounds or resume) and not base_dir:
        raise click.UsageError("--checkpoint-rounds and --resume need --base-dir")
    engine = FederatedEngine(cfg, comm=comm, base_dir=Path(base_dir) if base_dir else None)
    if resume:
        start = engine.resume()
        if comm.is_root:
            click.echo(json.dumps({"resumed_at_round": start}))
    for report in engine.run():
        if comm.is_root:
            line = {"round": report.round_id, "status": report.status,
                    "clients": report.num_clients, **report.agg_metrics,
                    "wall_s": round(report.wall_s, 4)}
            if phases:
                line["phases_ms"] = {k: round(v, 3) for k, v in report.phases_ms.items()}
            click.echo(json.dumps(line))
    if comm.is_root:
        click.echo(json.dumps({"eval": engine.evaluate()}))
    engine.close()
    comm.finish(0)  # the process group is destroyed before the interpreter shuts down


@main.command(context_settings={"ignore_unknown_options": True})
@click.argument("args", nargs=-1, type=click.UNPROCESSED)
def bench(args) -> None:
    import runpy
    import sys

    root = Path(__file__).resolve().parent.parent
    sys.argv = [str(root / "bench.py"), *args]
    runpy.run_path(str(root / "bench.py"), run_name="__main__")


@main.command()
@click.option("--host", default="127.0.0.1")
@click.option("--port", default=8080)
@click.option("--model", default="mnist_cnn")
@click.option("--rounds", default=2)
@click.option("--min-clients", default=2)
@click.option("--completion-rate", default=1.0)
@click.option("--timeout", default=300)
@click.option("--base-dir", default="runs/server", type=click.Path())
def serve(host, port, model, rounds, min_clients, completion_rate, timeout, base_dir) -> None:
    """HTTP server + coordinator for remote clients (the reference's deployment shape)."""
    from nanofed_amd import Coordinator, CoordinatorConfig, FedAvgAggregator, HTTPServer, ModelManager, coordinate
    from nanofed_amd.models import build_model

    async def go() -> None:
        server = HTTPServer(host, port)
        await server.start()
        try:
            coord = Coordinator(
                ModelManager(build_model(model)), FedAvgAggregator(), server,
                CoordinatorConfig(rounds, min_clients, completion_rate, timeout, Path(base_dir)),
            )
            await coordinate(coord)
        finally:
            await server.stop()

    asyncio.run(go())


if __name__ == "__main__":
    main()
