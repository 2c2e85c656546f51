"""bench.py's sequence with the wall time of every round of the second (bare) loop."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

import gc
_gc_t = [0.0]
def _gc_cb(phase, info):
    if phase == "start":
        _gc_t[0] = time.perf_counter()
    else:
        dt = (time.perf_counter() - _gc_t[0]) * 1000
        if dt > 1.0:
            print(f"gc gen{info['generation']} {dt:.1f} ms collected={info['collected']} tracked={len(gc.get_objects())}", flush=True)
gc.callbacks.append(_gc_cb)
clients = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.cuda.set_device(0)
comm = Communicator.from_env()
cfg = EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=7500, batch_size=64,
                   local_epochs=2, learning_rate=0.1, eval_samples=10000,
                   dataset="synthetic_mnist_hard")
eng = FederatedEngine(cfg, comm=comm, base_dir=tempfile.mkdtemp(prefix="stall_"))


def timed(n):
    comm.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(n):
        a = time.perf_counter()
        eng.run_round(flush=False, prefetch_next=i + 1 < n)
        marks.append(round((time.perf_counter() - a) * 1000, 2))
    a = time.perf_counter(); eng.flush(); fl = time.perf_counter() - a
    a = time.perf_counter(); torch.cuda.synchronize(); sy = time.perf_counter() - a
    comm.barrier(); torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("total ms", round(total * 1000, 1), "flush", round(fl * 1000, 2), "sync", round(sy * 1000, 2), marks, flush=True)


for _ in range(3):
    eng.run_round()
timed(20)
eng.evaluate()
eng.disable_persistence()
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
timed(10)
pr.disable()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(14)
print(buf.getvalue()[-3500:], flush=True)
timed(10)
eng.close()
