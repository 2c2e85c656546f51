"""Train-loss curve of the GPT-2 configuration for a few learning rates (the loss spikes of
plain SGD at 0.2 are a property of the configuration, not of the kernels)."""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchmarks.run_configs import engine_config  # noqa: E402
from nanofed_amd.engine import FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 14
for lr in [float(v) for v in sys.argv[2:]] or [0.2, 0.1]:
    comm = Communicator.from_env()
    cfg = dataclasses.replace(engine_config("gpt2_125m", 8, 0), learning_rate=lr)
    eng = FederatedEngine(cfg, comm=comm)
    losses = [round(eng.run_round().agg_metrics["loss"], 3) for _ in range(rounds)]
    eng.synchronize()
    print("lr", lr, "losses", losses, "eval", eng.evaluate(), flush=True)
    del eng
    comm.close()
    torch.cuda.empty_cache()
