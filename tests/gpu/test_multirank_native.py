"""Two ranks sharing the one GPU (gloo, host-staged collectives) must produce
the same global model as one rank hosting all clients: placement of clients on
ranks must not change the training of any client."""

import os
import subprocess
import sys

import pytest
import torch

from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CFG = dict(model="mnist_cnn", num_clients=4, samples_per_client=[320, 256, 192, 128],
           batch_size=64, local_epochs=2, eval_samples=256, backend="native", seed=5,
           placement_invariant=True)

WORKER = r"""
import sys, torch
sys.path.insert(0, {root!r})
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
comm = Communicator.from_env(backend="gloo", device="cuda:0")
eng = FederatedEngine(EngineConfig(**{cfg!r}), comm=comm)
reps = eng.run({rounds!r})
torch.cuda.synchronize()
if comm.is_root:
    torch.save({{"flat": eng.global_arena.data.cpu(), "loss": reps[-1].agg_metrics["loss"],
                "weights": [c["weight"] for c in reps[-1].client_metrics],
                "segments": sorted(eng._segment_graphs)}}, {out!r})
comm.finish(0)
"""


def _free_port() -> int:
    """A TCP port that is free right now (a fixed one can still be in TIME_WAIT from
    the previous parametrisation)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("nproc", [2, 3])
def test_ranks_sharing_the_gpu_give_the_bits_of_one_rank(tmp_path, nproc):
    """Placement-invariant mode: one kernel geometry whatever the number of clients per
    rank, client rows added in client order on every rank (ordered reduce).  Four clients
    on 2 ranks (2 + 2) and on 3 ranks (2 + 1 + 1), three rounds: the global model has the
    bits of the one-rank run."""
    out = tmp_path / "dist.pt"
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, cfg=CFG, out=str(out), rounds=3))
    proc = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
        capture_output=True, text=True, timeout=280,
        env=dict(os.environ, MASTER_ADDR="127.0.0.1"),
    )
    assert proc.returncode == 0, proc.stderr[-3000:]
    dist = torch.load(out, weights_only=True)
    torch.cuda.set_device(0)
    single = FederatedEngine(EngineConfig(**CFG), comm=Communicator(device=torch.device("cuda", 0)))
    reps = single.run(3)
    assert dist["weights"] == pytest.approx([320 / 896, 256 / 896, 192 / 896, 128 / 896])
    assert dist["segments"] == ["mid", "post", "pre", "screen"]
    assert torch.equal(dist["flat"], single.global_arena.data.cpu())
    assert dist["loss"] == reps[-1].agg_metrics["loss"]


RCCL_WORKER = r"""
import sys, torch
sys.path.insert(0, {root!r})
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
comm = Communicator.from_env()          # nccl (= RCCL) on cuda:0, group forced by the env
assert comm.backend == "nccl" and comm.group is not None and not comm._single
cfg = dict({cfg!r}, reduce_algo={algo!r})
eng = FederatedEngine(EngineConfig(**cfg), comm=comm)
reps = eng.run(4)
comm.barrier()
t = comm.max_over_ranks(1.5)
torch.cuda.synchronize()
torch.save({{"flat": eng.global_arena.data.cpu(), "loss": reps[-1].agg_metrics["loss"], "max": t,
            "segments": sorted(eng._segment_graphs)}}, {out!r})
comm.finish(0)
"""


@pytest.mark.parametrize("algo", ["ordered", "allreduce", "rs_ag"])
def test_rccl_collectives_single_rank_rehearsal(tmp_path, algo):
    """Every collective of the round goes through a real RCCL communicator (a
    group of one rank): broadcast, all-gather of the records, the arena reduce (all-to-all
    of row slices + all-gather for "ordered", all-reduce, or reduce-scatter + all-gather),
    barrier, max."""
    out = tmp_path / f"rccl_{algo}.pt"
    script = tmp_path / "worker.py"
    script.write_text(RCCL_WORKER.format(root=ROOT, cfg=CFG, out=str(out), algo=algo))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", NANOFED_FORCE_GROUP="1")
    proc = subprocess.run([sys.executable, str(script)], capture_output=True, text=True,
                          timeout=280, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    got = torch.load(out, weights_only=True)
    torch.cuda.set_device(0)
    single = FederatedEngine(EngineConfig(**CFG), comm=Communicator(device=torch.device("cuda", 0)))
    reps = single.run(4)
    assert got["max"] == 1.5
    # the device work between the eager collectives was replayed from graphs (recorded in
    # the second round, replayed in rounds 2-4)
    assert got["segments"] == (["mid", "post", "pre", "screen"] if algo == "ordered"
                               else ["post", "pre", "screen"])
    torch.testing.assert_close(got["flat"], single.global_arena.data.cpu(), rtol=0, atol=0)
    assert got["loss"] == pytest.approx(reps[-1].agg_metrics["loss"], rel=1e-6)


RESNET_CFG = dict(model="resnet18", dataset="synthetic_cifar10", num_clients=3,
                  samples_per_client=[96, 64, 64], batch_size=32, local_epochs=1, eval_samples=64,
                  learning_rate=0.05, backend="native", seed=7)


def test_grouped_resnet_two_ranks_track_one_rank(tmp_path):
    """The ResNet trainer that steps all clients of a rank per launch: clients 0-1 on one
    rank and client 2 on the other against all three in one group.  The group size picks
    the convolution tile (another summation order), so the models agree to bf16 noise,
    not bit for bit."""
    out = tmp_path / "dist.pt"
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, cfg=RESNET_CFG, out=str(out), rounds=1))
    proc = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
        capture_output=True, text=True, timeout=280, env=dict(os.environ, MASTER_ADDR="127.0.0.1"),
    )
    assert proc.returncode == 0, proc.stderr[-3000:]
    dist = torch.load(out, weights_only=True)
    torch.cuda.set_device(0)
    single = FederatedEngine(EngineConfig(**RESNET_CFG),
                             comm=Communicator(device=torch.device("cuda", 0)))
    reps = single.run(1)
    assert type(single.trainer).__name__ == "ResNetNativeTrainer"
    assert dist["weights"] == pytest.approx([96 / 224, 64 / 224, 64 / 224])
    ref = single.global_arena.data.cpu()
    rel = float((dist["flat"] - ref).norm() / ref.norm())
    assert rel < 0.02, rel
    assert dist["loss"] == pytest.approx(reps[-1].agg_metrics["loss"], rel=0.05)
