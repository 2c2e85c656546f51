import os, sys, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.engine.mnist_native import NativeBuffers
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
print("trainer _run_epoch", host_time(tr._run_epoch), flush=True)
streams = [torch.cuda.Stream() for _ in range(4)]
def conc(graphs):
    main = torch.cuda.current_stream()
    for s in streams: s.wait_stream(main)
    for i, g in enumerate(graphs):
        with torch.cuda.stream(streams[i]): g.replay()
    for s in streams: main.wait_stream(s)
print("trainer graphs, fresh streams", host_time(lambda: conc(tr._graphs)), flush=True)
# harness-style groups built now
pool, shards = tr.pool, tr.shards
graphs = []
for i in range(4):
    g0 = 2 * i
    b = NativeBuffers(pool.layout, 2, tr.device, tr.buf.images, tr.buf.labels, pool.data[g0:],
                      pool.layout.numel, pool.grad[g0:], tr.max_steps, 0.1307, 0.3081, cfg,
                      client_base=g0, train=True, seed=0, parent=tr.buf, first=g0)
    b.train_steps(0, 2); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        b.train_steps(0, tr.max_steps)
    graphs.append(g)
print("harness graphs", host_time(lambda: conc(graphs)), flush=True)
print("geometry trainer", [(g.ctx.split, g.ctx.WPC, g.ctx.conv1_split, g.ctx.fuse_sgd, g.ctx.ragged) for g in tr._groups], flush=True)
