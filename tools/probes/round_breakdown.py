"""Where a headline round's time goes on the main stream: events around every part of
MnistCnnNativeTrainer.train_round and of the engine's round, steady state."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

torch.cuda.set_device(0)
extra = {}
if "--dp" in sys.argv:  # the non-IID DP-SGD configuration of benchmarks/run_configs.py
    from nanofed_amd.engine import DPConfig

    extra = dict(partition="dirichlet", dirichlet_alpha=0.1,
                 dp=DPConfig(max_gradient_norm=1.0, noise_multiplier=1.1))
cfg = EngineConfig(model="mnist_cnn", num_clients=8, samples_per_client=7500, batch_size=64,
                   local_epochs=2, learning_rate=0.1, eval_samples=1000,
                   prefetch_rounds="--no-prefetch" not in sys.argv,
                   dataset="synthetic_mnist_hard", **extra)
eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
eng.run(3)
tr = eng.trainer
marks = []


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


orig_epoch, orig_take, orig_pack = tr._run_epoch, tr._take_indices, tr.buf.pack


def run_epoch():
    a = ev(); orig_epoch(); marks.append(("epoch", a, ev()))


def take(r, e):
    a = ev(); orig_take(r, e); marks.append(("take_indices", a, ev()))


tr._run_epoch, tr._take_indices = run_epoch, take
orig_train = tr.train_round


def train_round(r):
    a = ev(); out = orig_train(r); marks.append(("train_round", a, ev())); return out


tr.train_round = train_round
eng.synchronize()
t0 = time.perf_counter()
R = 10
eng.run(R)
eng.synchronize()
wall = (time.perf_counter() - t0) / R * 1000
tot = {}
for name, a, b in marks:
    tot.setdefault(name, []).append(a.elapsed_time(b))
out = {k: [round(sum(v) / len(v), 3), len(v)] for k, v in tot.items()}
# gaps between consecutive train_round spans = tail + host (or nothing when queued early)
spans = [(a, b) for n, a, b in marks if n == "train_round"]
gaps = [spans[i][1].elapsed_time(spans[i + 1][0]) for i in range(len(spans) - 1)]
print(json.dumps({"wall_ms_per_round": round(wall, 3), "mean_ms[count]": out,
                  "between_train_rounds_ms": round(sum(gaps) / len(gaps), 3),
                  "trial_epoch_ms": tr.stream_report}))
