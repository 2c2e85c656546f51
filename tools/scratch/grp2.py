import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.engine.mnist_native import NativeBuffers
from nanofed_amd.parallel import Communicator
torch.cuda.set_device(0)
comm = Communicator(device=torch.device("cuda", 0))
def timeit(fn, steps, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1000 / reps / steps, 1)
for samples in (640, 7500):
    cfg = EngineConfig(model="mnist_cnn", num_clients=8, samples_per_client=samples, batch_size=64,
                       local_epochs=1, eval_samples=64, backend="native", graph_capture=False)
    eng = FederatedEngine(cfg, comm=comm)
    tr = eng.trainer; pool, shards = tr.pool, tr.shards
    STEPS = tr.max_steps
    tr._fill_indices(0, 0); tr.buf.pack()
    out = {"steps": STEPS}
    for shared in (False, True):
        groups = []
        for i in range(4):
            g0 = 2 * i
            kw = dict(parent=tr.buf, first=g0) if shared else {}
            b = NativeBuffers(pool.layout, 2, tr.device, tr.buf.images, tr.buf.labels, pool.data[g0:],
                              pool.layout.numel, pool.grad[g0:], tr.max_steps, 0.1307, 0.3081, cfg,
                              client_base=g0, train=True, seed=0, **kw)
            if not shared:
                b.nvalid.copy_(tr.buf.nvalid[g0:g0+2]); b.idx.copy_(tr.buf.idx[g0:g0+2]); b.pack()
            b.train_steps(0, 2); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                b.train_steps(0, STEPS)
            groups.append((b, g))
        streams = [torch.cuda.Stream() for _ in range(4)]
        def conc():
            main = torch.cuda.current_stream()
            for s in streams: s.wait_stream(main)
            for i, (_, g) in enumerate(groups):
                with torch.cuda.stream(streams[i]): g.replay()
            for s in streams: main.wait_stream(s)
        out["shared" if shared else "own"] = timeit(conc, STEPS)
    g = torch.cuda.CUDAGraph()
    tr.buf.train_steps(0, 2); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        tr.buf.train_steps(0, STEPS)
    out["single"] = timeit(g.replay, STEPS)
    print(json.dumps(out), flush=True)
