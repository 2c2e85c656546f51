"""Run-to-run determinism of the DP-SGD configuration and of variants of it: two engines per
variant in one process, the global models compared bit for bit after a few rounds."""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.run_configs import engine_config  # noqa: E402
from nanofed_amd.engine import DPConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
base = engine_config("mnist_cnn_dp", 8, 0)
variants = {
    "dp dirichlet (config 3)": base,
    "dp iid": dataclasses.replace(base, partition="iid"),
    "no dp dirichlet": dataclasses.replace(base, dp=None),
    "dp sigma 0 dirichlet": dataclasses.replace(base, dp=DPConfig(max_gradient_norm=1.0, noise_multiplier=0.0)),
    "dp dirichlet eager": dataclasses.replace(base, graph_capture=False),
    "dp dirichlet 2 clients": dataclasses.replace(base, num_clients=2),
}
only = os.environ.get("VARIANT")
for name, cfg in variants.items():
    if only and only not in name:
        continue
    outs = []
    for i in range(2):
        comm = Communicator.from_env()
        eng = FederatedEngine(cfg, comm=comm)
        losses = [round(eng.run_round().agg_metrics["loss"], 6) for _ in range(rounds)]
        eng.synchronize()
        outs.append((eng.global_arena.data.clone(), losses))
        eng.close()
        comm.close()
    same = bool(torch.equal(outs[0][0], outs[1][0]))
    print("%-26s same bits: %-5s max diff %.3e  losses %s / %s" % (
        name, same, float((outs[0][0] - outs[1][0]).abs().max()), outs[0][1], outs[1][1]), flush=True)
