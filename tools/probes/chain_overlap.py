"""How much do the launch chains of the client groups slow each other down, kernel by kernel?

For every stage of the native MNIST step: a hipGraph of N back-to-back launches of that stage
per client group (pairs of clients), timed (a) for one group alone and (b) for all four groups
on the trainer's concurrent streams.  A ratio near 1 says the chains overlap freely for that
kernel, near 4 that they run one after the other.

    python tools/probes/chain_overlap.py [--clients 8] [--launches 40]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=8)
    ap.add_argument("--launches", type=int, default=40)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = FederatedEngine(
        EngineConfig(model="mnist_cnn", num_clients=args.clients, samples_per_client=64 * 16,
                     batch_size=64, local_epochs=1, eval_samples=64, backend="native"),
        comm=Communicator(device=dev))
    tr = eng.trainer
    eng.run_round()  # captures the epoch graphs and chooses the stream set
    torch.cuda.synchronize()
    groups, streams = tr._groups, tr._group_streams
    main_stream = torch.cuda.current_stream(dev)
    stages = ["conv1_fwd", "conv2_fwd", "fc1_fwd", "head", "fc1_dgrad", "bwd_pair", "conv2_wgrad",
              "finalize_sgd", "full_step"]
    out = {}
    for name in stages:
        graphs = []
        for grp in groups:
            def body(grp=grp):
                if name == "full_step":
                    grp.train_steps(0, max(1, args.launches // 8))
                else:
                    for _ in range(args.launches):
                        grp.run_stage(name, 0, 0)
            body()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            graphs.append(g)
        n = max(1, args.launches // 8) if name == "full_step" else args.launches

        def timed(which):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record(main_stream)
                for st in streams:
                    st.wait_stream(main_stream)
                for i in which:
                    with torch.cuda.stream(streams[i % len(streams)]):
                        graphs[i].replay()
                for st in streams:
                    main_stream.wait_stream(st)
                e1.record(main_stream)
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1000.0 / n)
            return best

        alone = timed([0])
        both = timed([0, 1]) if len(groups) > 1 else alone
        every = timed(list(range(len(groups))))
        out[name] = {"alone_us": round(alone, 2), "two_us": round(both, 2), "all_us": round(every, 2),
                     "ratio_all": round(every / alone, 2)}
        print(name, out[name], flush=True)
    print(json.dumps({"clients": args.clients, "groups": len(groups), "per_launch": out}))


if __name__ == "__main__":
    main()
