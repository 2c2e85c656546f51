"""Host-side pieces of the trainers that step all clients per launch: the grouping of
clients into launches, backend selection off the GPU, and the tile / split heuristics of
the batched GEMM."""

import torch

from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.engine.resnet_native import _runs, supports
from nanofed_amd.models import build_model
from nanofed_amd.ops import gemm_ops
from nanofed_amd.parallel import Communicator


def test_runs_of_consecutive_clients():
    assert _runs([]) == []
    assert _runs([0, 1, 2, 3]) == [(0, 4)]
    assert _runs([0, 2, 3, 5]) == [(0, 1), (2, 2), (5, 1)]
    # no run longer than the group limit
    assert _runs(list(range(7)), longest=3) == [(0, 3), (3, 3), (6, 1)]
    assert _runs([1, 2, 3, 4, 8, 9], longest=2) == [(1, 2), (3, 2), (8, 2)]


def test_grouped_resnet_trainer_is_not_offered_without_a_gpu():
    cfg = EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=2,
                       samples_per_client=32, batch_size=16)
    model = build_model("resnet18")
    assert not supports(cfg, model, torch.device("cpu"))
    # on a GPU: width, batch size, DP and dtype decide
    dev = torch.device("cuda", 0)
    narrow = build_model("resnet18", width=16)
    assert not supports(cfg, narrow, dev)
    odd = EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=2,
                       samples_per_client=32, batch_size=12)
    assert not supports(odd, model, dev)
    fp32 = EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=2,
                        samples_per_client=32, batch_size=16, dtype="fp32")
    assert not supports(fp32, model, dev)


def test_cpu_engine_keeps_the_per_client_trainers():
    eng = FederatedEngine(
        EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=2,
                     samples_per_client=16, batch_size=8, eval_samples=8,
                     model_kwargs={"width": 8}, local_epochs=1),
        comm=Communicator(device=torch.device("cpu")),
    )
    assert type(eng.trainer).__name__ == "TorchGroupTrainer"
    rep = eng.run_round()
    assert rep.status == "COMPLETED"
    # the BatchNorm step counters are kept in the arena rows: global value + the round's steps
    assert int(eng.global_state_dict()["bn1.num_batches_tracked"]) == 2


def test_batched_gemm_heuristics_count_the_tiles_of_all_problems():
    # one 4096 x 768 x 768 weight gradient needs split-K (36 tiles), eight of them do not
    assert gemm_ops.choose_splits(768, 768, 4096, gemm_ops.GEO_128_2) > 1
    assert gemm_ops.choose_splits(768, 768, 4096, gemm_ops.GEO_128_2, batch=8) == 1
    # a tiny output stays split even in a batch
    assert gemm_ops.choose_splits(64, 32, 65536, gemm_ops.GEO_128_2, batch=8) > 1
    # a round and a half of 256 x 256 workgroups with a very long reduction is cut in two
    # (three full rounds); with a short reduction the slabs would cost more than the tail
    assert gemm_ops.choose_splits(4096, 768, 50257, gemm_ops.GEO_256, batch=8) == 2
    assert gemm_ops.choose_splits(4096, 768, 3072, gemm_ops.GEO_256, batch=8) == 1
    assert gemm_ops.choose_splits(4096, 3072, 50257, gemm_ops.GEO_256, batch=8) == 1  # six full rounds
    # geometry: the tile count that selects the wide tiles is the batch's
    one = gemm_ops.choose_geometry(4096, 2304, 768, gemm_ops.NT)
    many = gemm_ops.choose_geometry(4096, 2304, 768, gemm_ops.NT, batch=8)
    assert one == gemm_ops.GEO_128x192 and many == gemm_ops.GEO_128x192
    # 6.0 rounds of 256 x 256 tiles: the big tile; 4.5 rounds (above): 128 x 192
    assert gemm_ops.choose_geometry(4096, 3072, 768, gemm_ops.NT, batch=8) == gemm_ops.GEO_256
    assert gemm_ops.choose_geometry(4096, 3072, 768, gemm_ops.NN, batch=8) == gemm_ops.GEO_256
    # a group's data gradient with a long reduction takes the big tile, a shorter one not
    assert gemm_ops.choose_geometry(4096, 768, 3072, gemm_ops.NN, batch=8) == gemm_ops.GEO_256
    assert gemm_ops.choose_geometry(4096, 768, 2304, gemm_ops.NN, batch=8) == gemm_ops.GEO_128_2
    assert gemm_ops.choose_geometry(4096, 768, 3072, gemm_ops.NN) == gemm_ops.GEO_128_2
    # a group's weight gradient: short side first -> 256 x 128, long side first -> 128 x 128
    assert gemm_ops.choose_geometry(768, 3072, 4096, gemm_ops.TN, batch=8) == gemm_ops.GEO_256x128
    assert gemm_ops.choose_geometry(3072, 768, 4096, gemm_ops.TN, batch=8) == gemm_ops.GEO_128_2
