"""Native bf16 against torch fp32 curves without dropout: per-round numbers side by side."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

lr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
dataset = sys.argv[2] if len(sys.argv) > 2 else "synthetic_mnist_hard"
CFG = dict(model="mnist_cnn", dataset=dataset, num_clients=4, samples_per_client=4096,
           batch_size=64, local_epochs=2, learning_rate=lr, eval_samples=2000, seed=5,
           model_kwargs={"dropout": False})
for kw in (dict(), dict(backend="torch", dtype="fp32", client_streams=False),
           dict(backend="torch", dtype="bf16", client_streams=False)):
    eng = FederatedEngine(EngineConfig(**CFG, **kw), comm=Communicator(device=torch.device("cuda", 0)))
    rows = []
    for _ in range(3):
        rep = eng.run_round()
        ev = eng.evaluate()
        rows.append([round(v, 4) for v in (rep.agg_metrics["loss"], rep.agg_metrics["accuracy"],
                                           ev["loss"], ev["accuracy"])])
    print(eng.backend, kw.get("dtype", "bf16"), rows, flush=True)
    eng.close()
