import os, sys, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
torch.cuda.set_device(0)
comm = Communicator(device=torch.device("cuda", 0))
for streams in (True, False):
    cfg = EngineConfig(model="mnist_cnn", num_clients=8, samples_per_client=7500, batch_size=64,
                       local_epochs=2, eval_samples=64, backend="native", client_streams=streams)
    eng = FederatedEngine(cfg, comm=comm)
    eng.run_round(); eng.run_round()
    tr = eng.trainer
    out = {"groups": len(tr._groups)}
    for name, fn in (("run_epoch", tr._run_epoch), ("train_round", lambda: tr.train_round(5)), ("engine_round", eng.run_round)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = round((time.perf_counter() - t0) / 4 * 1e3, 2)
    print(json.dumps(out), flush=True)
