"""Federated ResNet-18 on CIFAR-10 (binary batches) or CIFAR-10-shaped synthetic data.

    python examples/resnet_cifar/run_federated.py --clients 8 --rounds 3
    python examples/resnet_cifar/run_federated.py --data-dir ~/data --samples-per-client 6250
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29521 examples/resnet_cifar/run_federated.py --clients 8

On a MI355X the engine picks ``ResNetNativeTrainer``: explicit forward and backward on
hand-written kernels with all clients of the GPU in every launch (implicit-GEMM MFMA
convolutions, BatchNorm + residual + ReLU, a two-launch head, one SGD launch), the step
replayed from a hipGraph; FedAvg is one fused reduce (+ an RCCL all-reduce when several
GPUs take part).  BatchNorm running statistics are averaged with the weights, integer
buffers stay out of the float average.  ``--data-dir`` names a folder that holds
``cifar-10-batches-bin/`` or ``cifar-10-binary.tar.gz`` (the pickled "python version" is
never read); without it the hard synthetic set is used (four patterns per class, 8 %
random labels).  ``NANOFED_RESNET_NATIVE=0`` trains client by
client through autograd instead.
"""

from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--samples-per-client", type=int, default=1024)
    ap.add_argument("--local-epochs", type=int, default=1)
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--alpha", type=float, default=0.0, help="> 0: Dirichlet(alpha) non-IID shards")
    ap.add_argument("--base-dir", default=None)
    ap.add_argument("--data-dir", default=None, help="folder with the CIFAR-10 binary batches")
    args = ap.parse_args()
    comm = Communicator.from_env()
    cfg = EngineConfig(
        model="resnet18", dataset="cifar10" if args.data_dir else "synthetic_cifar10_hard",
        data_dir=os.path.expanduser(args.data_dir or "./data"), num_clients=args.clients,
        samples_per_client=args.samples_per_client, batch_size=64,
        local_epochs=args.local_epochs, learning_rate=args.lr, eval_samples=2000,
        partition="dirichlet" if args.alpha > 0 else "iid", dirichlet_alpha=args.alpha or 0.1,
    )
    engine = FederatedEngine(cfg, comm=comm, base_dir=args.base_dir)
    for report in engine.run(args.rounds):
        if comm.is_root:
            print(json.dumps({"round": report.round_id, "status": report.status,
                              **report.agg_metrics, "wall_s": round(report.wall_s, 3)}))
    if comm.is_root:
        print(json.dumps({"eval": engine.evaluate()}))
    engine.close()
    comm.finish()  # the process group is destroyed before the interpreter shuts down


if __name__ == "__main__":
    main()
