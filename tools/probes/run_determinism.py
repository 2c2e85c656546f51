"""Two engines per benchmark configuration in one process: the global models after a few
rounds compared bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.run_configs import engine_config  # noqa: E402
from nanofed_amd.engine import FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

for name in sys.argv[1:] or ["mnist_cnn", "mnist_cnn_dp", "resnet18", "gpt2_125m"]:
    outs = []
    for i in range(2):
        comm = Communicator.from_env()
        eng = FederatedEngine(engine_config(name, 8, 0), comm=comm)
        losses = [round(eng.run_round().agg_metrics["loss"], 6) for _ in range(3)]
        eng.synchronize()
        outs.append((eng.global_arena.data.clone(), losses))
        eng.close()
        comm.close()
        del eng
        torch.cuda.empty_cache()
    print("%-14s same bits: %-5s max diff %.3e  losses %s" % (
        name, bool(torch.equal(outs[0][0], outs[1][0])),
        float((outs[0][0] - outs[1][0]).abs().max()), outs[0][1]), flush=True)
