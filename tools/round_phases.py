"""Per-phase timing of federated rounds on the native MNIST path.
    python tools/round_phases.py [--clients 8] [--rounds 6]
Prints the mean of the engine's HIP-event phase timings and the wall time."""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--samples", type=int, default=7500)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    comm = Communicator(device=torch.device("cuda", 0))
    cfg = EngineConfig(model="mnist_cnn", num_clients=args.clients, samples_per_client=args.samples,
                       batch_size=64, local_epochs=2, learning_rate=0.1, eval_samples=1000)
    eng = FederatedEngine(cfg, comm=comm)
    eng.run(2)
    eng.synchronize()
    t0 = time.perf_counter()
    reps = eng.run(args.rounds)
    eng.synchronize()
    wall = (time.perf_counter() - t0) / args.rounds * 1000
    keys = reps[0].phases_ms.keys()
    mean = {k: round(sum(r.phases_ms[k] for r in reps) / len(reps), 3) for k in keys}
    steps = 2 * ((args.samples + 63) // 64)
    print(json.dumps({"clients": args.clients, "wall_ms_per_round": round(wall, 3),
                      "phases_ms": mean, "steps_per_round": steps,
                      "host_wall_s": [round(r.wall_s * 1000, 3) for r in reps]}))


if __name__ == "__main__":
    main()
