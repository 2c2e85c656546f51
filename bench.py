"""Headline benchmark: federated rounds per second, MNIST-CNN FedAvg, 8 clients.

One "step" is one complete federated round: broadcast of the global model,
``local_epochs`` epochs of SGD (batch 64) on every client's 7,500-sample IID
shard, screening, weighted FedAvg reduce (RCCL all-reduce across GPUs), new
global model, and - as in the reference's round
(nanofed/orchestration/coordinator.py:346-357,247-280) - a saved model version
(``.pt`` + config JSON) and a round-metrics JSON, written by rank 0 into a
scratch directory; the writes of the last round are flushed before the clock
stops.  ``rounds_per_s_no_persistence`` is the same loop timed once more without
the writes.  The 8 clients are spread over the N GPUs (8/N per GPU), so the total
work is fixed as N grows (strong scaling).  Data is synthetic MNIST-shaped
(``synthetic_mnist_hard``: three patterns per class, pixel noise, 5 % random labels -
held-out accuracy cannot exceed about 0.95 and takes many rounds to get there, so the
accuracy printed next to the rate shows whether the rounds still train), weights are
random-init.

Reference number: 53.48 s per round = 0.0187 rounds/s (3 clients, CPU, logged
in examples/mnist/tutorial.ipynb:523 of the reference).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8
"""

from __future__ import annotations

import argparse
import gc
import json
import os
import shutil
import sys
import tempfile
import time

# the host driver of these machines supports dmabuf IPC only: without this, RCCL's
# peer-memory exchange between the ranks of a node fails (hipIpcGetMemHandle)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

BASELINE_ROUNDS_PER_S = 1.0 / 53.48


def _restore_stdout(saved_fd: int) -> None:
    import ctypes

    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)  # whatever C code still holds in its stdio buffers
    except OSError:
        pass
    os.dup2(saved_fd, 1)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5, help="timed federated rounds")
    ap.add_argument("--warmup", type=int, default=1, help="untimed rounds")
    ap.add_argument("--clients", type=int, default=8)
    ap.add_argument("--local-epochs", type=int, default=2)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--samples-per-client", type=int, default=7500)
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--backend", default="auto", choices=["auto", "native", "torch"])
    ap.add_argument("--model", default="mnist_cnn")
    ap.add_argument("--dataset", default="synthetic_mnist_hard",
                    choices=["synthetic_mnist_hard", "synthetic_mnist", "mnist"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="run on CPU/gloo (plumbing check)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not queue the next round before the host has finished the current one")
    ap.add_argument("--verbose", action="store_true", help="keep the INFO/DEBUG log lines")
    ap.add_argument("--no-persist", action="store_true",
                    help="leave the per-round model version / metrics files out of the timed rounds")
    ap.add_argument("--eval-samples", type=int, default=10000)
    ap.add_argument("--dump-model", default=None,
                    help="rank 0 saves the final global model (flat fp32 tensor) to this file")
    args = ap.parse_args()

    from nanofed_amd.engine import EngineConfig, FederatedEngine
    from nanofed_amd.parallel import Communicator

    # Libraries print to stdout on their own (RCCL's version banner, gloo's
    # connection notes).  Until the result line is due, file descriptor 1 points at
    # stderr, so stdout carries exactly one JSON line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if not args.verbose:
        # three log lines per saved round are formatted on the writer thread otherwise
        from nanofed_amd.utils import LogConfig, Logger, LogLevel

        Logger().configure(LogConfig(level=LogLevel.WARNING))
    # keep the collective library quiet on stdout (rank 0 prints exactly one JSON line)
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        print(f"WORLD_SIZE={world} does not match --gpus {args.gpus}", file=sys.stderr)
        return 2
    comm = Communicator.from_env(
        backend="gloo" if args.cpu else None, device="cpu" if args.cpu else None
    )
    cfg = EngineConfig(
        model=args.model,
        dataset=args.dataset,
        num_clients=args.clients,
        local_epochs=args.local_epochs,
        batch_size=args.batch_size,
        learning_rate=args.lr,
        samples_per_client=args.samples_per_client,
        dtype=args.dtype,
        backend=args.backend,
        graph_capture=not args.no_graph,
        eval_samples=args.eval_samples,
        prefetch_rounds=not args.no_prefetch,
    )
    scratch = None
    if not args.no_persist and comm.is_root:
        scratch = tempfile.mkdtemp(prefix="nanofed_bench_")
    engine = FederatedEngine(cfg, comm=comm, base_dir=scratch)

    def sync() -> None:
        if comm.device.type == "cuda":
            torch.cuda.synchronize(comm.device)

    def timed(rounds: int) -> tuple[float, list]:
        comm.barrier()
        sync()
        t0 = time.perf_counter()
        # the engine queues round i + 1 on the device while the host finishes round i
        # (as FederatedEngine.run does); the last timed round queues nothing
        reps = [engine.run_round(flush=False, prefetch_next=i + 1 < rounds) for i in range(rounds)]
        engine.flush()  # the last rounds' files are on disk before the clock stops
        sync()
        comm.barrier()
        sync()
        return comm.max_over_ranks(time.perf_counter() - t0), reps

    for _ in range(args.warmup):
        engine.run_round()
    # everything built so far (modules, graphs, buffers) lives as long as the process: keep
    # it out of the cyclic collector's scans, whose full passes take 50-60 ms here
    gc.collect()
    gc.freeze()
    elapsed, reports = timed(args.steps)
    final = engine.evaluate()
    dump = engine.global_arena.data.detach().cpu().clone() if args.dump_model else None

    # the same loop without the per-round files (what the writer thread costs)
    bare = None
    if scratch is not None or (not args.no_persist and comm.world_size > 1):
        engine.disable_persistence()
        bare, _ = timed(min(args.steps, 10))
    if comm.is_root:
        rounds_per_s = args.steps / elapsed
        last = reports[-1] if reports else None
        line = {
            "metric": f"federated rounds/sec (MNIST-CNN FedAvg, {args.clients} clients)",
            "value": rounds_per_s,
            "unit": "rounds/s",
            "n_gpus": comm.world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": rounds_per_s / BASELINE_ROUNDS_PER_S,
            "dtype": args.dtype,
            "data": f"synthetic MNIST-shaped ({args.dataset}), IID shards, random-init weights",
            "persistence": "off" if args.no_persist else
            "model version (.pt + JSON) and metrics JSON per round, writer thread, flushed in the timed region",
            "rounds_per_s_no_persistence": (min(args.steps, 10) / bare) if bare else None,
            "stream_set": (last.trainer_info or None) if last else None,
            "final_test_accuracy": final.get("accuracy"),
            "final_train_accuracy": last.agg_metrics["accuracy"] if last else None,
            "final_train_loss": last.agg_metrics["loss"] if last else None,
            "samples_per_s": rounds_per_s
            * args.clients * args.samples_per_client * args.local_epochs,
            "config": {
                "model": args.model,
                "clients": args.clients,
                "local_epochs": args.local_epochs,
                "global_batch": args.batch_size * args.clients,
                "batch_size": args.batch_size,
                "samples_per_client": args.samples_per_client,
                "seq_len": None,
                "optimizer": f"SGD lr={args.lr}",
                "backend": type(engine.trainer).__name__,
                "parallelism": f"fed{args.clients}/gpu{comm.world_size}",
            },
        }
        if dump is not None:
            torch.save(dump, args.dump_model)
        _restore_stdout(real_stdout)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    engine.close()
    if scratch is not None:
        shutil.rmtree(scratch, ignore_errors=True)
    comm.finish(0)  # multi-process: leaves here, past the libraries' exit handlers
    return 0


if __name__ == "__main__":
    sys.exit(main())
