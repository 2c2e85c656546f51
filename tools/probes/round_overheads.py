"""Phase timings and host-side wall time of rounds at small client counts."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
from nanofed_amd.utils import Logger
from nanofed_amd.utils.logger import LogConfig, LogLevel
Logger().configure(LogConfig(level=LogLevel.WARNING))
comm = Communicator(device=torch.device("cuda", 0))
for g in (1, 2, 8):
    eng = FederatedEngine(EngineConfig(model="mnist_cnn", dataset="synthetic_mnist_hard", num_clients=g,
                                       samples_per_client=7500, local_epochs=2, eval_samples=2000), comm=comm)
    for _ in range(3):
        eng.run_round()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = [eng.run_round() for _ in range(10)]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    ph = {k: round(sum(r.phases_ms.get(k, 0) for r in reps) / 10, 3) for k in reps[-1].phases_ms}
    print(g, "clients: wall", round(wall, 3), "ms/round; phases", ph, "sum", round(sum(ph.values()), 3), flush=True)
    eng.close()
