import os, sys, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
torch.cuda.set_device(0)
comm = Communicator(device=torch.device("cuda", 0))
for n in (8, 4):
    cfg = EngineConfig(model="mnist_cnn", num_clients=n, samples_per_client=7500, batch_size=64,
                       local_epochs=2, eval_samples=64, backend="native")
    eng = FederatedEngine(cfg, comm=comm)
    for _ in range(3): eng.run_round()
    print(n, "trials ms", [round(t, 2) for t, _ in eng.trainer._stream_trials], flush=True)
