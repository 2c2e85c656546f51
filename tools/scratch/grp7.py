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
def host_time(fn, reps=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return round(1e3 * (time.perf_counter() - t0) / reps, 2)
def conc(graphs, streams):
    main = torch.cuda.current_stream()
    for s in streams: s.wait_stream(main)
    for i, g in enumerate(graphs):
        with torch.cuda.stream(streams[i]): g.replay()
    for s in streams: main.wait_stream(s)
print("A trainer _run_epoch", host_time(tr._run_epoch), flush=True)
print("B conc(trainer graphs, trainer streams)", host_time(lambda: conc(tr._graphs, tr._group_streams)), flush=True)
fresh = [torch.cuda.Stream() for _ in range(4)]
print("C conc(trainer graphs, fresh streams)", host_time(lambda: conc(tr._graphs, fresh)), flush=True)
old = tr._group_streams; tr._group_streams = fresh
print("D trainer _run_epoch with fresh streams", host_time(tr._run_epoch), flush=True)
tr._group_streams = old
print("E trainer _run_epoch again", host_time(tr._run_epoch), flush=True)
print("ids", [s.cuda_stream for s in old], [s.cuda_stream for s in fresh], "idx", tr._idx_stream.cuda_stream, flush=True)
