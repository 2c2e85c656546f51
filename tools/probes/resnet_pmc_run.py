"""A few eager (not graph-captured) steps of the grouped ResNet-18 trainer at the
benchmark shape (8 clients x 64 images), as the program of tools/pmc_passes.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

torch.cuda.set_device(0)
cfg = EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=8,
                   samples_per_client=192, batch_size=64, local_epochs=1, learning_rate=0.05,
                   eval_samples=64, graph_capture=False)
eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
rep = eng.run_round()
torch.cuda.synchronize()
print(type(eng.trainer).__name__, rep.status, rep.agg_metrics)
