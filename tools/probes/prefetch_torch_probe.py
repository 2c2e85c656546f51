"""Is the PyTorch-op trainer bit-reproducible run to run on the GPU, with and without the
early queueing of the next round?"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator


def go(prefetch, rounds=4, **kw):
    torch.cuda.set_device(0)
    cfg = dict(model="mnist_cnn", num_clients=4, samples_per_client=640, batch_size=64,
               local_epochs=2, eval_samples=256, backend="torch", prefetch_rounds=prefetch)
    cfg.update(kw)
    eng = FederatedEngine(EngineConfig(**cfg), comm=Communicator(device=torch.device("cuda", 0)))
    reps = eng.run(rounds)
    return eng.global_arena.data.clone(), [r.agg_metrics["loss"] for r in reps]


for label, kw in (("default", {}), ("no client streams", {"client_streams": False}),
                  ("no graph", {"graph_capture": False})):
    a, la = go(False, **kw)
    b, lb = go(False, **kw)
    c, lc = go(True, **kw)
    print(label, "plain==plain", torch.equal(a, b), "plain==early", torch.equal(a, c),
          "max diff", float((a - c).abs().max()), flush=True)
    print("  losses", [round(x, 5) for x in la], [round(x, 5) for x in lb], [round(x, 5) for x in lc], flush=True)
