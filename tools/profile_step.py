"""A few eager optimisation steps of the native MNIST trainer (for rocprofv3).
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python tools/profile_step.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

clients = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
eng = FederatedEngine(
    EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=64 * steps,
                 batch_size=64, local_epochs=1, eval_samples=64, backend="native",
                 graph_capture=False),
    comm=Communicator(device=torch.device("cuda", 0)))
tr = eng.trainer
tr._fill_indices(0, 0)
tr.buf.pack()
tr.buf.train_steps(0, steps)
torch.cuda.synchronize()
print("done")
