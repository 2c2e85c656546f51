"""Which tensors differ between a run with an injected failure + rollback and a run without."""
import pathlib
import sys
import tempfile

import torch

from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.parallel import Communicator


def run(fail, base, backend):
    torch.cuda.set_device(0)
    cfg = EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=3,
                       samples_per_client=64, batch_size=32, local_epochs=1, eval_samples=64,
                       learning_rate=0.05, seed=9, backend=backend, checkpoint_rounds=True,
                       max_update_norm=1000.0)
    faults = FaultPlan(drop={0: {2}}, poison={1: {0}}, fail={2: "HIP error: injected"} if fail else {})
    eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)), base_dir=base,
                          faults=faults)
    reps = eng.run(4)
    eng.flush()
    return eng, reps


for backend in sys.argv[1:] or ["native", "torch"]:
    t = pathlib.Path(tempfile.mkdtemp())
    a, ra = run(False, t / "a", backend)
    b, rb = run(True, t / "b", backend)
    print(backend, [r.round_id for r in rb], b.recoveries,
          "equal" if torch.equal(a.global_arena.data, b.global_arena.data) else "DIFFERENT", flush=True)
    for x, y in zip(ra, rb):
        print("  round", x.round_id, x.agg_metrics["loss"], y.agg_metrics["loss"], x.num_clients, y.num_clients)
    sa, sb = a.global_arena.views(), b.global_arena.views()
    n = 0
    for k in sa:
        d = float((sa[k].float() - sb[k].float()).abs().max())
        if d > 0 and n < 8:
            print("   ", k, d)
            n += 1
