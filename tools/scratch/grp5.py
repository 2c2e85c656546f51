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
t0 = time.perf_counter()
for i in range(4): tr._run_epoch()
torch.cuda.synchronize()
print(os.environ.get("NF_SKIP"), os.environ.get("NF_STRIDE"), f"epoch {1e3*(time.perf_counter()-t0)/4:.2f} ms", [s.cuda_stream % 100000 for s in tr._group_streams], flush=True)
