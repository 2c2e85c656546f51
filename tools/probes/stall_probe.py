"""Wall time of single rounds around evaluate() / disable_persistence() calls."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

clients = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.cuda.set_device(0)
cfg = EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=7500, batch_size=64,
                   local_epochs=2, learning_rate=0.1, eval_samples=10000,
                   dataset="synthetic_mnist_hard")
eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)),
                      base_dir=tempfile.mkdtemp(prefix="stall_"))


def rounds(n, label):
    out = []
    for i in range(n):
        t0 = time.perf_counter()
        eng.run_round(flush=False, prefetch_next=i + 1 < n)
        out.append(round((time.perf_counter() - t0) * 1000, 2))
    eng.flush()
    eng.synchronize()
    print(label, out, flush=True)


rounds(6, "warm")
rounds(6, "steady")
t0 = time.perf_counter(); eng.evaluate(); eng.synchronize()
print("evaluate ms", round((time.perf_counter() - t0) * 1000, 2), flush=True)
rounds(6, "after evaluate")
t0 = time.perf_counter(); eng.evaluate(); eng.synchronize()
print("evaluate ms", round((time.perf_counter() - t0) * 1000, 2), flush=True)
rounds(6, "after 2nd evaluate")
eng.disable_persistence()
rounds(6, "after disable_persistence")
rounds(6, "again")
eng.close()
