"""Per-kernel timing of the native MNIST-CNN step (hipEvent brackets around
back-to-back launches of one stage).  Usage:
    python tools/bench_mnist_stages.py [--clients 8] [--reps 200]
"""

from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nanofed_amd.engine import DPConfig, EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.engine.mnist_native import STAGES  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=8)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only", default="", help="comma-separated stage names")
    ap.add_argument("--dp", action="store_true", help="per-sample DP-SGD step (two backward passes)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    comm = Communicator(device=torch.device("cuda", 0))
    cfg = EngineConfig(model="mnist_cnn", num_clients=args.clients, samples_per_client=640,
                       batch_size=64, local_epochs=1, eval_samples=64, backend="native",
                       graph_capture=False,
                       dp=DPConfig(max_gradient_norm=1.0, noise_multiplier=1.1) if args.dp else None)
    eng = FederatedEngine(cfg, comm=comm)
    tr = eng.trainer
    buf = tr.buf
    tr._fill_indices(0, 0)
    buf.pack()
    buf.train_steps(0, 2)  # realistic activations / gradients in every buffer
    torch.cuda.synchronize()
    out = {}
    # (stages that refuse the mode: the shared backward launches are for plain SGD, dp_pair
    # belongs to the norm pass - the context of a DP trainer is set up for the clipped pass)
    skip = (("pack", "finalize_sgd", "bwd_pair", "bwd_triple", "dp_pair") if args.dp
            else ("pack", "dp_scale", "dp_pair", "dp_clip"))
    order = [n for n in STAGES if n not in skip]
    if args.only:
        order = args.only.split(",")
    per_graph = 25

    def timed_graph(fn, launches):
        """hipGraph of `launches` back-to-back calls: no host launch gaps."""
        fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(launches):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        reps = max(1, args.reps // launches)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000.0 / (reps * launches)

    for name in order:
        out[name] = timed_graph(lambda: buf.run_stage(name, 0, 0), per_graph)
    if not args.only:
        out["full_step"] = timed_graph(lambda: buf.train_steps(0, 1), 8)
    out["sum_of_stages"] = sum(out[n] for n in order)
    print(json.dumps({"clients": args.clients, "us": {k: round(v, 2) for k, v in out.items()}}))


if __name__ == "__main__":
    main()
