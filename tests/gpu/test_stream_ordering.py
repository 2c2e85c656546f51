"""Stream-ordering check of the multi-stream paths: a run with every side stream
collapsed onto the main stream (``NANOFED_SERIALIZE_STREAMS=1``: client chains, index
prefetch, snapshot copy, GPT-2 lanes all serial) must produce the same bits as the
normal, overlapping run.  A missing event or wait shows up as a difference."""

import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKER = r"""
import sys, torch
sys.path.insert(0, {root!r})
from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.parallel import Communicator
cfg = EngineConfig(**{cfg!r})
eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)), base_dir={base!r},
                      faults=FaultPlan(drop={{2: {{1}}}}))
reps = eng.run(4)
eng.close()
torch.save({{"flat": eng.global_arena.data.cpu(), "loss": [r.agg_metrics["loss"] for r in reps],
            "streams": len(getattr(eng.trainer, "_group_streams", []))}}, {out!r})
"""


def _run(tmp_path, name, cfg, serialize):
    out = tmp_path / f"{name}.pt"
    script = tmp_path / f"{name}.py"
    script.write_text(WORKER.format(root=ROOT, cfg=cfg, base=str(tmp_path / name), out=str(out)))
    env = dict(os.environ, NANOFED_SERIALIZE_STREAMS="1" if serialize else "0")
    proc = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True,
                          timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    return torch.load(out, weights_only=True)


@pytest.mark.parametrize("clients", [2, 8])
def test_overlapping_and_serialized_runs_agree_bit_for_bit(tmp_path, clients):
    cfg = dict(model="mnist_cnn", num_clients=clients, samples_per_client=[448, 320] * (clients // 2),
               local_epochs=2, eval_samples=128, seed=9)
    normal = _run(tmp_path, "normal", cfg, serialize=False)
    serial = _run(tmp_path, "serial", cfg, serialize=True)
    assert torch.equal(normal["flat"], serial["flat"])
    assert normal["loss"] == serial["loss"]
