import os, sys, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
torch.cuda.set_device(0)
comm = Communicator(device=torch.device("cuda", 0))
cfg = EngineConfig(model="mnist_cnn", num_clients=8, samples_per_client=7500, batch_size=64,
                   local_epochs=2, eval_samples=64, backend="native", client_streams=True)
eng = FederatedEngine(cfg, comm=comm)
eng.run_round(); eng.run_round()
tr = eng.trainer
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter(); tr._run_epoch(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"grouped epoch: enqueue {1e3*(t1-t0):.2f} ms, drain {1e3*(t2-t1):.2f} ms", flush=True)
# one group's graph alone
g = tr._graphs[0]
for i in range(2):
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"one group graph: enqueue {1e3*(t1-t0):.2f} ms, drain {1e3*(t2-t1):.2f} ms", flush=True)
