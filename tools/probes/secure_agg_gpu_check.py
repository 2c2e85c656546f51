"""Pairwise-masked rounds against plain rounds on the GPU (native and torch backends)."""
import torch
from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.parallel import Communicator
from nanofed_amd.utils import Logger
from nanofed_amd.utils.logger import LogConfig, LogLevel

Logger().configure(LogConfig(level=LogLevel.WARNING))
comm = Communicator(device=torch.device("cuda", 0))
for backend in ("native", "torch"):
    base = dict(model="mnist_cnn", num_clients=4, samples_per_client=512, local_epochs=1, backend=backend)
    for drop in ({}, {0: {2}}):
        a = FederatedEngine(EngineConfig(**base), comm=comm, faults=FaultPlan(drop=dict(drop)))
        b = FederatedEngine(EngineConfig(secure_agg="pairwise", **base), comm=comm, faults=FaultPlan(drop=dict(drop)))
        a2 = FederatedEngine(EngineConfig(**base), comm=comm, faults=FaultPlan(drop=dict(drop)))
        for r in range(2):
            a.run_round(); b.run_round(); a2.run_round()
            d = (a.global_arena.data - b.global_arena.data).abs()
            d2 = (a.global_arena.data - a2.global_arena.data).abs()
            print(backend, "drop" if drop else "full", "round", r, "pairwise-plain", float(d.max()), "at", int(d.argmax()),
                  "plain-plain", float(d2.max()), flush=True)
        for e in (a, b, a2):
            e.close()
