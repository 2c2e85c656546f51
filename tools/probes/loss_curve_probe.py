"""Print 3-round curves of the native and the fp32 torch trainer for a few seeds."""
import torch
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
from nanofed_amd.utils import Logger
from nanofed_amd.utils.logger import LogConfig, LogLevel

Logger().configure(LogConfig(level=LogLevel.WARNING))
comm = Communicator(device=torch.device("cuda", 0))
for n in (2048, 4096):
    for seed in (5, 6):
        for kw in (dict(), dict(backend="torch", dtype="fp32")):
            cfg = EngineConfig(model="mnist_cnn", dataset="synthetic_mnist_hard", num_clients=4,
                               samples_per_client=n, batch_size=64, local_epochs=2, learning_rate=0.1,
                               eval_samples=2000, seed=seed, **kw)
            eng = FederatedEngine(cfg, comm=comm)
            out = []
            for _ in range(3):
                rep = eng.run_round()
                ev = eng.evaluate()
                out.append((round(rep.agg_metrics["loss"], 3), round(rep.agg_metrics["accuracy"], 3),
                            round(ev["loss"], 3), round(ev["accuracy"], 3)))
            print(n, seed, eng.backend, out, flush=True)
            eng.close()
